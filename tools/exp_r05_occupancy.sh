#!/bin/bash
# round 5: encoder throughput by resident workgroups per CU — bash tools/exp_r05_occupancy.sh "variant ..." "blocks ..."
cd $GRAFT_REPO_ROOT
for v in $1; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  [ "$v" = "product" ] && unset CJ_HIP_LIB
  for l in $2; do
    CJ_ENC_BLOCKS=$l python bench.py --op compress --codec lz4 --no-cpu-baseline --traffic off --steps 3 --warmup 1 --experiment-no-verify 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v blocks/CU $l: %.1f GB/s' % d['value'])"
  done
done
