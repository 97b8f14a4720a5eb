# fused kernel vs parse + decode at batch sizes around CJ_FUSED_MAX_CHUNKS:  bash tools/exp_fused_crossover.sh [codec]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
C=${1:-lz4}
for N in ${SIZES:-4096 8192 12288 16384 20000 24000 28000}; do
  for F in 1 0; do
    echo -n "chunks $N  CJ_FUSED=$F  "
    CJ_FUSED=$F python bench.py --codec $C --chunks $N --unique 2048 $BENCH_ARGS --no-cpu-baseline --traffic off --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GB/s %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
  done
done
