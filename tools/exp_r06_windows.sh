# round 6: batches of small chunks on windows of their size (CJ_FLAG_CHUNKS_LE_32K / _16K) — tests, then rates against the 64 KiB window
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_small_windows_gpu.py -x -q -m gpu 2>&1 | tail -6
run() {
  echo "== $@"
  timeout 600 python bench.py --no-cpu-baseline --traffic off --steps 10 "$@" 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GB/s %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
  tail -1 /tmp/err.txt | cut -c1-200
}
for C in lz4 snappy; do
run --codec $C --chunk-bytes 32768 --chunks 200000
run --codec $C --chunk-bytes 16384 --chunks 400000
run --codec $C --chunk-bytes 8192 --chunks 800000
run --codec $C --chunk-bytes 32768 --chunks 16384
run --codec $C --chunk-bytes 16384 --chunks 16384
run --codec $C --chunk-bytes 16384 --chunks 4096
done
run --data corpus64k --chunk-bytes 32768 --chunks 200000
run --data corpus64k --chunk-bytes 16384 --chunks 400000
run
