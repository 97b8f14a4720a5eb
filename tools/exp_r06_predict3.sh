# round 6, predictor part 3: 32 KiB windows on 32 KiB chunks, four wavefronts per workgroup, by workgroups (= chains) per CU
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() {
  V=$1; shift
  L=$PWD/cramjam_amd/variants/libcramjam_hip_$V.so; [ "$V" = base ] && L=$PWD/cramjam_amd/libcramjam_hip.so
  echo "== $V $@"
  CJ_HIP_LIB=$L timeout 600 python bench.py --no-cpu-baseline --traffic off --steps 10 --phase-profile "$@" 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GB/s %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
  grep -i "cycles/chunk" /tmp/err.txt | head -1 | cut -c1-140
}
for V in w32k2 w32k3 w32k4 w32k2x8 w32k4x5; do
run $V --chunk-bytes 32768 --chunks 200000
done
for V in w32k2 w32k3 w32k4 w32k4x5; do
run $V --data corpus64k --chunk-bytes 32768 --chunks 200000
done
