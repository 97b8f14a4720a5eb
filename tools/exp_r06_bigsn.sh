# f06b: big_elem's Snappy straight-line shapes (literal headers up to 4 bytes, copy-4, literal behind literal): the walk of configs[4]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-product bigsn product bigsn}; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  [ "$v" = "product" ] && unset CJ_HIP_LIB
  [ "$v" != "product" ] && echo "$v tests: $(timeout 900 python -m pytest tests/test_big_chunks_gpu.py -x -q 2>&1 | tail -1)"
  for args in "--codec snappy --chunk-bytes 262144 --chunks 8192 --unique 2048 --steps 20" "--chunk-bytes 262144 --chunks 8192 --unique 2048 --steps 20" "--workload mixed256k --steps 20"; do
    python bench.py $args --no-cpu-baseline --traffic off 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v [$args]: %.1f GB/s %.3f ms' % (d['value'], d['ms_per_step']))"
  done
done
