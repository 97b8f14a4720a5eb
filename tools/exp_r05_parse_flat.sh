#!/bin/bash
# round 5: the parse kernels' refill loads as global_load instead of flat_load (lane_stream.hpp), and __builtin_amdgcn_ballot_w64 for every ballot
cd $GRAFT_REPO_ROOT
for v in before product balb before product balb; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  [ "$v" = "product" ] && unset CJ_HIP_LIB
  for args in "" "--codec snappy" "--data corpus64k --steps 20"; do
    python bench.py --no-cpu-baseline --traffic off $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v [$args] %.1f GB/s %.3f ms' % (d['value'], d['ms_per_step']))"
  done
done
CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_balb.so timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
