# bench line + phase profile for named library variants (tools/build_variant.sh):  bash tools/exp_variants_quick.sh base prioA ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for V in "$@"; do
  L=$PWD/cramjam_amd/variants/libcramjam_hip_$V.so; [ "$V" = base ] && L=$PWD/cramjam_amd/libcramjam_hip.so
  echo "== $V"
  CJ_HIP_LIB=$L python bench.py --no-cpu-baseline --traffic off --steps 20 --phase-profile $BENCH_ARGS 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GB/s %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
  grep -i "cycles/chunk" /tmp/err.txt | cut -c1-110
done
