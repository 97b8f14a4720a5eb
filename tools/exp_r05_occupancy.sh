cd $GRAFT_REPO_ROOT
for l in 2 3 5 7 9; do
  export CJ_ENC_LDS_BLOCKS=$l CJ_ENC_TABLE_BLOCKS=0
  python bench.py --op compress --codec lz4 --no-cpu-baseline --traffic off --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lds_blocks/CU $l: %.1f GB/s' % d['value'])"
done
