# e31: long matches finished by four groups of sixteen lanes (256 bytes per head and round trip) instead of the whole wavefront head by head;
# lane trips before that 8 / 4 / 2
cd $GRAFT_REPO_ROOT
for V in ${VARIANTS:-product}; do
  [ $V = product ] && unset CJ_HIP_LIB || export CJ_HIP_LIB=$PWD/cramjam_amd/variants/libcramjam_hip_$V.so
  echo "== $V: $(timeout 900 python -m pytest tests/test_enc2_gpu.py -x -q 2>&1 | tail -1)  $(N=1500 timeout 600 python tests/perf/fuzz_enc2.py 2>&1 | tail -1)"
  for F in alice29.txt mr kppkn.gtb geo.protodata xml html_x_4; do
    CJ_CORPUS_FILES=$F python bench.py --op compress --data corpus64k --codec lz4 --chunks 20000 --no-cpu-baseline --traffic off --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %-14s %7.1f GB/s' % ('$F', d['value']))"
  done
  for C in lz4 snappy; do
  python bench.py --op compress --codec $C --no-cpu-baseline --traffic off --steps 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   synth $C %.1f GB/s' % d['value'])"
  python bench.py --op compress --data corpus64k --codec $C --chunks 20000 --no-cpu-baseline --traffic off --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   corpus $C %.1f GB/s' % d['value'])"
  done
done
