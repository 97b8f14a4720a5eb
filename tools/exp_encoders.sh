cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
for v in "$@"; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  [ "$v" = "product" ] && unset CJ_HIP_LIB
  for codec in lz4 snappy; do
    timeout 300 python bench.py --op compress --codec $codec --no-cpu-baseline --traffic off --steps 5 --warmup 2 > $O/${v}_$codec.json 2> $O/${v}_$codec.err
    python -c "
import json,sys
t=open('$O/${v}_$codec.json').read().strip()
if not t: print('$v $codec: no JSON', open('$O/${v}_$codec.err').read().strip().splitlines()[-1][:200]); sys.exit(0)
d=json.loads(t.splitlines()[-1]); print('$v $codec value %.1f GB/s ms/step %.3f ratio %s' % (d['value'], d['ms_per_step'], d['config'].get('ratio')))" | tee -a $O/summary.txt
  done
done
