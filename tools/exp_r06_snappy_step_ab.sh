# f06 A/B: the Snappy straight-line step before / after (variants stepold / stepnew), three rounds
cd $GRAFT_REPO_ROOT
for round in ${ROUNDS:-1 2 3}; do
for v in ${VARIANTS:-stepold stepnew}; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  for args in "--codec snappy --chunks 8192 --unique 2048" "--data corpus64k --codec snappy --chunks 8192" "--codec snappy --chunk-bytes 16384 --chunks 4096 --unique 2048"; do
    python bench.py $args --no-cpu-baseline --traffic off --steps 40 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v [$args]: %.1f GB/s %.3f ms' % (d['value'], d['ms_per_step']))"
  done
done
done
