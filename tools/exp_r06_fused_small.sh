# f05: the one-kernel path on 32 / 16 KiB windows (batches of small chunks) against parse kernel + small-window decoder, by batch size
cd $GRAFT_REPO_ROOT
echo "tests: $(timeout 1800 python -m pytest tests/test_small_windows_gpu.py tests/test_gpu_parity.py tests/test_corpus_gpu.py -x -q 2>&1 | tail -1)"
for C in ${CODECS:-lz4 snappy}; do
for S in 32768 16384 8192; do
for N in ${SIZES:-512 1024 2048 4096 8192 16384 32768 65536}; do
  for P in fused kernel; do
    echo -n "$C ${S} B x $N  --parse $P  "
    python bench.py --codec $C --chunk-bytes $S --chunks $N --unique 2048 --parse $P --no-cpu-baseline --traffic off --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GB/s %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
  done
done
done
done
