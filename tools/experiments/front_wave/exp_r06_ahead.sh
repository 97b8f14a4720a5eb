# round 6, a01: the front wavefront (kAhead) — S0 + D1 off the chunk's chain.  base = -DCJ_L2_AHEAD=0, product = 8 groups per lane, aN = N groups
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() {
  V=$1; shift
  L=$PWD/cramjam_amd/variants/libcramjam_hip_$V.so; [ "$V" = prod ] && L=$PWD/cramjam_amd/libcramjam_hip.so
  echo "== $V $@"
  CJ_HIP_LIB=$L timeout 600 python bench.py --no-cpu-baseline --traffic off --steps 20 --phase-profile "$@" 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GB/s %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
  grep -i "cycles/chunk" /tmp/err.txt | cut -c1-200 | uniq
}
for v in "$@"; do run $v; done
