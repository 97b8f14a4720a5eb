# d10: D3 — wavefronts whose batch lies more than N batches behind the completed ones sleep between polls (-DCJ_D3_FAR=N -DCJ_D3_FAR_SLEEP=k)
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-front32 front8 front32p1}; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  [ "$v" = "product" ] && unset CJ_HIP_LIB
  [ "$v" != "product" ] && echo "$v: $(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -1)"
  for args in "" "--data corpus64k --steps 20"; do
  python bench.py $args --no-cpu-baseline --traffic off 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v [$args]: %.1f GB/s %.3f ms' % (d['value'], d['ms_per_step']))"
  done
  for F in mr alice29.txt; do CJ_CORPUS_FILES=$F python bench.py --data corpus64k --chunks 20000 --no-cpu-baseline --traffic off --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v %-14s %7.1f GB/s' % ('$F', d['value']))"; done
done
