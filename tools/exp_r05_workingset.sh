cd $GRAFT_REPO_ROOT
export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_w1.so
for cb in 8192 16384 32768 65536; do
  n=$((6553600000 / cb))
  python bench.py --op compress --codec lz4 --chunk-bytes $cb --chunks $n --no-cpu-baseline --traffic on --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['roofline']['traffic_detail']['kernels']; k=[x for x in t if 'encode' in x][0]
print('chunk %d: %.1f GB/s ratio %s hbm read %.1f GB write %.1f GB per step (input %.2f GB)' % ($cb, d['value'], d['config']['ratio'], t[k]['hbm_read_bytes']/1e9, t[k]['hbm_write_bytes']/1e9, $n*$cb/1e9))"
done
