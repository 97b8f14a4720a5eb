# g08: S0 of the next chunk inside D4 of the current one (LDS-DMA refill of the streamed-out slots): -DCJ_S0_PIPE=1
cd $GRAFT_REPO_ROOT
V=${V:-s0pipe}
export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$V.so
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_corpus_gpu.py -x -q 2>&1 | tail -2
for v in product $V product $V; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  [ "$v" = "product" ] && unset CJ_HIP_LIB
  for args in "" "--codec snappy" "--data corpus64k --steps 20"; do
  python bench.py $args --no-cpu-baseline --traffic off 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v [$args]: %.1f GB/s %.3f ms' % (d['value'], d['ms_per_step']))"
  done
done
python bench.py --phase-profile --no-cpu-baseline --traffic off --steps 5 2>&1 | grep "cycles/chunk"
