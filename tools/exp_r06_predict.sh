# round 6, predictor for the half-window decoder: (a) product on 64 KiB and on 32 KiB chunks, (b) four workgroups of four wavefronts per CU on
# 32 KiB windows (32 KiB chunks: what four chains in flight buy without the cross copies), (c) 256 threads at two workgroups per CU (the verdict's predictor)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() {
  V=$1; shift
  L=$PWD/cramjam_amd/variants/libcramjam_hip_$V.so; [ "$V" = base ] && L=$PWD/cramjam_amd/libcramjam_hip.so
  echo "== $V $@"
  CJ_HIP_LIB=$L timeout 600 python bench.py --no-cpu-baseline --traffic off --steps 20 --phase-profile "$@" 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GB/s %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
  grep -i "cycles/chunk" /tmp/err.txt | cut -c1-140
  tail -3 /tmp/err.txt | cut -c1-200
}
run base
run base --chunk-bytes 32768 --chunks 200000
run w32k4 --chunk-bytes 32768 --chunks 200000
run t256
run base --data corpus64k
run t256 --data corpus64k
