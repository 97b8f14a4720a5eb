cd $GRAFT_REPO_ROOT
for v in product cap4096 cap2048 product cap4096 cap2048; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  [ "$v" = "product" ] && unset CJ_HIP_LIB
  for args in "--workload mixed256k --chunks 16384" "--codec lz4 --chunk-bytes 262144 --chunks 8192 --unique 2048"; do
  python bench.py $args --no-cpu-baseline --traffic off --steps 10 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v [$args]: %.1f GB/s %.3f ms' % (d['value'], d['ms_per_step']))"
  done
done
