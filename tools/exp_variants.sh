#!/bin/bash
# On the GPU box: bench + phase profile for each tuning variant given as NAME arguments (built by tools/build_variant.sh).
# usage: bash tools/exp_variants.sh TAG name1 name2 ... ; env BENCH_ARGS adds bench.py arguments
cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for v in "$@"; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  [ "$v" = "product" ] && unset CJ_HIP_LIB
  echo "== $v" | tee -a $O/summary.txt
  timeout 300 python bench.py --phase-profile --no-cpu-baseline --steps 10 $BENCH_ARGS > $O/$v.json 2> $O/$v.err
  grep "cycles/chunk" $O/$v.err | cut -c1-200 | tee -a $O/summary.txt
  python -c "
import json,sys
t=open('$O/$v.json').read().strip()
if not t: print('no JSON line (verification failed or crashed):', open('$O/$v.err').read().strip().splitlines()[-1][:150]); sys.exit(0)
d=json.loads(t.splitlines()[-1]); print('value %.1f GB/s  ms/step %.3f  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))" 2>&1 | tee -a $O/summary.txt
done
