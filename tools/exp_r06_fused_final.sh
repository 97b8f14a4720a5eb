# f04 on the product build: all GPU tests, the fallback paths forced by variant builds (work list of 256 groups: walked P4; 40 rows: walked P3 + P4),
# and the fused / parse-kernel crossover (bench.py --parse fused | kernel)
cd $GRAFT_REPO_ROOT
echo "product: $(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -1)"
for v in flistg flistr; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  echo "$v: $(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_corpus_gpu.py tests/test_frames_gpu.py -x -q -k 'not lists_its_walks' 2>&1 | tail -1)"
done
unset CJ_HIP_LIB
for C in lz4 snappy; do
for N in 8192 16384 20000 24000 28000 32768 40000; do
  for P in fused kernel; do
    echo -n "$C chunks $N  --parse $P  "
    python bench.py --codec $C --chunks $N --unique 2048 --parse $P --no-cpu-baseline --traffic off --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GB/s %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
  done
done
done
for C in lz4 snappy; do
for N in 16384 24000 32768; do
  for P in fused kernel; do
    echo -n "corpus $C chunks $N  --parse $P  "
    python bench.py --data corpus64k --codec $C --chunks $N --parse $P --no-cpu-baseline --traffic off --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GB/s %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
  done
done
done
