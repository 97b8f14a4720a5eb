# g07: the LZ4 record expansion (D1) with ONE dependent LDS read per sequence (the offset field read as 8 bytes brings the next token): -DCJ_LZ4_D1_FAST=1
cd $GRAFT_REPO_ROOT
export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_d1fast.so
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_corpus_gpu.py tests/test_frames_gpu.py tests/test_large_gpu.py -x -q 2>&1 | tail -2
for v in product d1fast product d1fast; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  [ "$v" = "product" ] && unset CJ_HIP_LIB
  for args in "" "--data corpus64k --steps 20"; do
  python bench.py $args --no-cpu-baseline --traffic off 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v [$args]: %.1f GB/s %.3f ms' % (d['value'], d['ms_per_step']))"
  done
done
