# rates of the paths that run through the slab / linked modes of the workgroup decoder, for library variants: bash tools/exp_slab_paths.sh base prev
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
for V in "$@"; do
  L=$PWD/cramjam_amd/variants/libcramjam_hip_$V.so; [ "$V" = base ] && L=$PWD/cramjam_amd/libcramjam_hip.so
  echo "== $V"
  CJ_HIP_LIB=$L python bench.py --workload mixed256k --no-cpu-baseline --traffic off --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mixed256k GB/s %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
  CJ_HIP_LIB=$L MB=64 python tests/perf/single_buffer_rates.py 2>/dev/null | grep -i "decompress"
  CJ_HIP_LIB=$L python tests/perf/linked_frame_rate.py 2>/dev/null | tail -3
done
