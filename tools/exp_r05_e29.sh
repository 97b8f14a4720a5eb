cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_enc2_gpu.py tests/test_gpu_parity.py tests/test_large_gpu.py -x -q 2>&1 | tail -3
N=3000 timeout 900 python tests/perf/fuzz_enc2.py 2>&1 | tail -2
for F in mr kppkn.gtb geo.protodata xml nci html_x_4; do
    CJ_CORPUS_FILES=$F python bench.py --op compress --data corpus64k --codec lz4 --chunks 20000 --no-cpu-baseline --traffic off --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-14s %7.1f GB/s' % ('$F', d['value']))"
done
python bench.py --op compress --codec lz4 --no-cpu-baseline --traffic off --steps 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('synth %.1f GB/s' % d['value'])"
