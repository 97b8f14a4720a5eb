# round 6: small-chunk batches of medium size — parse inside the decoder (64 KiB windows) against parse kernel + small-window decoder
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() {
  echo -n "$@:  "
  timeout 600 python bench.py --no-cpu-baseline --traffic off --steps 30 "$@" 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GB/s %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
}
for S in 32768 16384; do for N in 1024 2048 4096 8192 16384 32768; do for P in fused kernel; do
run --chunk-bytes $S --chunks $N --unique 1024 --parse $P
done; done; done
