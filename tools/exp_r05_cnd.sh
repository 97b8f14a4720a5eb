#!/bin/bash
# round 5: does the e64 encoding of v_cndmask (vcc) pay in the real kernels?  product vs cnd64 variant: decode / compress lines + tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
PROBE_FROM=35 timeout 300 tools/_bin/issue_rate_probe | tee $O/issue_rate_probe_v3.txt
for v in product cnd64 product cnd64; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  [ "$v" = "product" ] && unset CJ_HIP_LIB
  for args in "" "--codec snappy" "--op compress" "--data corpus64k --steps 20"; do
    timeout 300 python bench.py --no-cpu-baseline --traffic off $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v [$args] %.1f GB/s %.3f ms' % (d['value'], d['ms_per_step']))" | tee -a $O/summary.txt
  done
done
export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_cnd64.so
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee -a $O/summary.txt
