# round 6, predictor part 2: the corpus as 32 KiB chunks — product (two workgroups of eight wavefronts) against four workgroups of four wavefronts on 32 KiB windows
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() {
  V=$1; shift
  L=$PWD/cramjam_amd/variants/libcramjam_hip_$V.so; [ "$V" = base ] && L=$PWD/cramjam_amd/libcramjam_hip.so
  echo "== $V $@"
  CJ_HIP_LIB=$L timeout 600 python bench.py --no-cpu-baseline --traffic off --steps 10 --phase-profile "$@" 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GB/s %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
  grep -i "cycles/chunk" /tmp/err.txt | head -1 | cut -c1-140
  tail -2 /tmp/err.txt | cut -c1-200
}
run base --data corpus64k --chunk-bytes 32768 --chunks 200000
run w32k4 --data corpus64k --chunk-bytes 32768 --chunks 200000
run base --codec snappy
run base --codec snappy --chunk-bytes 32768 --chunks 200000
run w32k4 --codec snappy --chunk-bytes 32768 --chunks 200000
