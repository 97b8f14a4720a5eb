# s08: the two barriers at the top of a chunk as LDS-only barriers (no wait for the previous chunk's streamed-out bytes)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_topbar.so
echo "topbar tests: $(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_corpus_gpu.py tests/test_small_windows_gpu.py tests/test_frames_gpu.py -x -q 2>&1 | tail -1)"
for round in 1 2 3; do
for v in product topbar; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  [ "$v" = "product" ] && unset CJ_HIP_LIB
  for args in "" "--codec snappy" "--chunks 8192 --unique 2048" "--chunk-bytes 16384 --chunks 400000" "--chunk-bytes 16384 --chunks 4096 --unique 2048" "--data corpus64k --steps 10"; do
    python bench.py $args --no-cpu-baseline --traffic off 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v [$args]: %.1f GB/s %.3f ms' % (d['value'], d['ms_per_step']))"
  done
done
done
for v in product topbar; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  [ "$v" = "product" ] && unset CJ_HIP_LIB
  echo "$v: $(python bench.py --no-cpu-baseline --traffic off --steps 10 --phase-profile 2>&1 | grep 'LDS decoder cycles' | tail -1)"
done
