# quick A/B on the GPU box: bench lines (+ phase profile) for the environment settings given as arguments, one per run
#   bash tools/exp_r04_quick.sh "CJ_SEG_KLOG=1" "CJ_PARSE=old" ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for E in "$@"; do
  echo "== $E"
  env $E python bench.py --no-cpu-baseline --traffic off --steps 20 --phase-profile 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GB/s %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
  grep -i "cycles/chunk" /tmp/err.txt | cut -c1-120
done
