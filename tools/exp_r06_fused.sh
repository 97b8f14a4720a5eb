# f04: the fused kernel's parse with LISTED walks (P3 = a subtraction, P4 = a copy of the list) against the three walks
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-flist0 flist flist0 flist}; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  if [ -z "$SKIP_TESTS" ]; then echo "$v: $(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_corpus_gpu.py tests/test_frames_gpu.py -x -q 2>&1 | tail -1)"; fi
  for args in "--chunks 1024 --unique 1024" "--chunks 8192 --unique 2048" "--chunks 16384 --unique 2048" "--codec snappy --chunks 1024 --unique 1024" "--codec snappy --chunks 8192 --unique 2048" "--data corpus64k --chunks 8192" "--data corpus64k --codec snappy --chunks 8192"; do
  python bench.py $args --no-cpu-baseline --traffic off --steps 30 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v [$args]: %.1f GB/s %.3f ms' % (d['value'], d['ms_per_step']))"
  done
  python bench.py --chunks 8192 --unique 2048 --no-cpu-baseline --traffic off --steps 10 --phase-profile 2>&1 | grep "LDS decoder cycles" | tail -1
done
