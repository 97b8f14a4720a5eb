#!/bin/bash
# round 5: the round-based encoders (cj_enc2.hpp) against the model, against the round-4 matcher (CJ_ENC_V1=1), with counters
# usage (on the GPU box): bash tools/exp_r05_enc.sh TAG [tests] [bench] [probe] [pmc]
cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for what in "$@"; do
case $what in
tests)
  timeout 900 python -m pytest tests/test_enc2_gpu.py -x -q 2>&1 | tail -15 | tee $O/pytest_enc2.txt
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_api_gpu.py -x -q -k "encode or encoder or compress or roundtrip" 2>&1 | tail -8 | tee $O/pytest_encode_parity.txt
  ;;
bench)
  for v1 in 0 1; do for codec in lz4 snappy; do
    CJ_ENC_V1=$v1 timeout 300 python bench.py --op compress --codec $codec --no-cpu-baseline --traffic off --steps 5 --warmup 2 > $O/bench_v1_${v1}_$codec.json 2> $O/bench_v1_${v1}_$codec.err
    python -c "
import json,sys
t=open('$O/bench_v1_${v1}_$codec.json').read().strip()
if not t: print('v1=$v1 $codec: no JSON', open('$O/bench_v1_${v1}_$codec.err').read().strip().splitlines()[-1][:300]); sys.exit(0)
d=json.loads(t.splitlines()[-1]); print('v1=$v1 $codec value %.1f GB/s ms/step %.3f ratio %s' % (d['value'], d['ms_per_step'], d['config'].get('ratio')))" | tee -a $O/summary.txt
  done; done
  timeout 300 python bench.py --op roundtrip --codec snappy --no-cpu-baseline --traffic off --steps 5 --warmup 2 2>/dev/null | tail -1 | tee $O/bench_roundtrip_snappy.json
  ;;
probe)
  timeout 600 tools/_bin/issue_rate_probe | tee $O/issue_rate_probe.txt
  ;;
pmc)
  bash tools/pmc_encode.sh $TAG product
  ;;
esac
done
