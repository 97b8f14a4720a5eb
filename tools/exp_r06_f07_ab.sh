# f07 A/B on the large batches: 1 / 8 / 32 claim counters
cd $GRAFT_REPO_ROOT
for round in ${ROUNDS:-1 2 3}; do
for v in ${VARIANTS:-claim1 claim8 product}; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  [ "$v" = "product" ] && unset CJ_HIP_LIB
  for args in "" "--codec snappy" "--chunk-bytes 16384 --chunks 400000" "--data corpus64k --steps 10"; do
    python bench.py $args --no-cpu-baseline --traffic off 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v [$args]: %.1f GB/s %.3f ms' % (d['value'], d['ms_per_step']))"
  done
done
done
