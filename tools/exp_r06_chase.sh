# e11: the selection's fallback as next-pointers + a readlane chase (variants chaseN = N parallel passes first; walk16 = the serial walk)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-walk16 chase16 chase6 chase3 walk16 chase16 chase6 chase3}; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  if [ -z "$SKIP_TESTS" ]; then echo "$v model identity: $(timeout 1200 python -m pytest tests/test_enc2_gpu.py -x -q -m gpu 2>&1 | tail -1)"; fi
  for args in "--op compress" "--op compress --codec snappy" "--op compress --data corpus64k" "--op compress --data corpus64k --codec snappy"; do
    timeout 600 python bench.py --no-cpu-baseline --traffic off --steps 8 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v [$args]: GB/s %.1f  ms/step %.3f  ratio %s' % (d['value'], d['ms_per_step'], d['config'].get('ratio')))"
  done
done
