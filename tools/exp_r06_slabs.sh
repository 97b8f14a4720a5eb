# round 6: the big-chunk slab decoder on 32 KiB slabs (four workgroups of four wavefronts per CU) against 64 KiB slabs (variant slab64)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_big_chunks_gpu.py -x -q -m gpu 2>&1 | tail -5
run() {
  V=$1; shift
  L=$PWD/cramjam_amd/variants/libcramjam_hip_$V.so; [ "$V" = base ] && L=$PWD/cramjam_amd/libcramjam_hip.so
  echo "== $V $@"
  CJ_HIP_LIB=$L timeout 600 python bench.py --no-cpu-baseline --traffic off --steps 10 "$@" 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GB/s %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
  tail -2 /tmp/err.txt | cut -c1-200
}
for V in base slab64; do
run $V --workload mixed256k
run $V --workload mixed256k --chunks 49152
run $V --chunk-bytes 262144 --chunks 8192
run $V --chunk-bytes 262144 --chunks 8192 --codec snappy
done
