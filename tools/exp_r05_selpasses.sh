cd $GRAFT_REPO_ROOT
for v in product sp16 sp32 sp64; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  [ "$v" = "product" ] && unset CJ_HIP_LIB
  for F in mr kppkn.gtb alice29.txt; do
    CJ_CORPUS_FILES=$F python bench.py --op compress --data corpus64k --codec lz4 --chunks 20000 --no-cpu-baseline --traffic off --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v %-14s %7.1f GB/s' % ('$F', d['value']))"
  done
  python bench.py --op compress --codec lz4 --no-cpu-baseline --traffic off --steps 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v synth %.1f GB/s' % d['value'])"
done
