# f08: the listed cells as aligned 16-byte cells (110 rows) against the 12-byte cells (144 rows)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_cell16.so
echo "cell16 tests: $(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_corpus_gpu.py tests/test_small_windows_gpu.py -x -q 2>&1 | tail -1)"
for round in 1 2 3; do
for v in cell12 cell16; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  for args in "--chunks 8192 --unique 2048" "--chunks 1024 --unique 1024" "--codec snappy --chunks 8192 --unique 2048" "--chunk-bytes 16384 --chunks 4096 --unique 2048" "--data corpus64k --chunks 8192"; do
    python bench.py $args --no-cpu-baseline --traffic off --steps 30 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v [$args]: %.1f GB/s %.3f ms' % (d['value'], d['ms_per_step']))"
  done
done
done
for v in cell12 cell16; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  echo "$v: $(python bench.py --chunks 8192 --unique 2048 --no-cpu-baseline --traffic off --steps 10 --phase-profile 2>&1 | grep 'LDS decoder cycles' | tail -1)"
done
