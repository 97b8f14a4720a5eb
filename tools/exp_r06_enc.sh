# round 6: the encoders with block-wise insertion (a position sees a table at most 128 positions stale; no coverage bitmap, no late insert):
# kernels against the model, round trips through the decoders, bench lines (synth and corpus, both codecs)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_enc2_gpu.py -x -q -m gpu 2>&1 | tail -6
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_api_gpu.py tests/test_large_gpu.py tests/test_frames_gpu.py -x -q -m gpu 2>&1 | tail -4
run() {
  echo "== $@"
  timeout 600 python bench.py --no-cpu-baseline --traffic off --steps 10 "$@" 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GB/s %.1f  ms/step %.3f  ratio %s' % (d['value'], d['ms_per_step'], d['config'].get('ratio')))"
  tail -1 /tmp/err.txt | cut -c1-200
}
run --op compress
run --op compress --codec snappy
run --op compress --data corpus64k
run --op compress --data corpus64k --codec snappy
run --op roundtrip --codec snappy
