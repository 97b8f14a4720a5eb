# d12: s_sleep in the resolver's empty poll trips — for the OTHER workgroup's latency-bound phases (the one-kernel path's parse)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for round in 1 2; do
for v in sleep0 sleep1 sleep3; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  for args in "--chunks 8192 --unique 2048" "--codec snappy --chunks 8192 --unique 2048" "--chunk-bytes 16384 --chunks 4096 --unique 2048" "--data corpus64k --chunks 8192" ""; do
    python bench.py $args --no-cpu-baseline --traffic off --steps 30 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v [$args]: %.1f GB/s %.3f ms' % (d['value'], d['ms_per_step']))"
  done
done
done
for v in sleep0 sleep1 sleep3; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  echo "$v: $(python bench.py --chunks 8192 --unique 2048 --no-cpu-baseline --traffic off --steps 10 --phase-profile 2>&1 | grep 'LDS decoder cycles' | tail -1)"
done
