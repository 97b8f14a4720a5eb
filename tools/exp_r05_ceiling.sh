#!/bin/bash
# round 5, x01: where is the encoder's shared ceiling?  The same kernel with the emission / the literal copies / the measuring pass's loads
# compiled out (outputs are wrong by construction: --experiment-no-verify), at 3 / 6 / 9 blocks per CU.
cd $GRAFT_REPO_ROOT
for v in product x_noemit x_nolit x_noext; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  [ "$v" = "product" ] && unset CJ_HIP_LIB
  python bench.py --op compress --codec lz4 --no-cpu-baseline --traffic off --steps 3 --warmup 1 --experiment-no-verify 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v: %.1f GB/s %.3f ms' % (d['value'], d['ms_per_step']))"
done
