# round 6, predictor part 4: smaller windows still — 16 KiB chunks on 16 KiB windows (8 workgroups of 2 wavefronts per CU), 8 KiB on 8 KiB (16 of 1)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() {
  V=$1; shift
  L=$PWD/cramjam_amd/variants/libcramjam_hip_$V.so; [ "$V" = base ] && L=$PWD/cramjam_amd/libcramjam_hip.so
  echo "== $V $@"
  CJ_HIP_LIB=$L timeout 600 python bench.py --no-cpu-baseline --traffic off --steps 10 --phase-profile "$@" 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GB/s %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
  grep -i "cycles/chunk" /tmp/err.txt | head -1 | cut -c1-140
  tail -1 /tmp/err.txt | cut -c1-200
}
run nofwd --chunk-bytes 16384 --chunks 400000
run w16k4 --chunk-bytes 16384 --chunks 400000
run w16k6 --chunk-bytes 16384 --chunks 400000
run w16k8 --chunk-bytes 16384 --chunks 400000
run w8k16 --chunk-bytes 8192 --chunks 800000
run w16k8 --data corpus64k --chunk-bytes 16384 --chunks 400000
run w32k4nf --data corpus64k --chunk-bytes 32768 --chunks 200000
run nofwd --data corpus64k
