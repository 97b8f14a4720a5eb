# bench line + HBM traffic for named library variants (tools/build_variant.sh):  bash tools/exp_variants_traffic.sh base s0nt ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for V in "$@"; do
  L=$PWD/cramjam_amd/variants/libcramjam_hip_$V.so; [ "$V" = base ] && L=$PWD/cramjam_amd/libcramjam_hip.so
  echo "== $V"
  CJ_HIP_LIB=$L python bench.py --no-cpu-baseline --traffic on --steps 20 $BENCH_ARGS 2>/tmp/err.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('GB/s %.1f  ms/step %.3f  traffic %.2f GB' % (d['value'], d['ms_per_step'], (r.get('traffic') or 0)/1e9))
for k,v in (r.get('traffic_detail') or {}).get('kernels',{}).items(): print('   %-50s read %.2f write %.2f GB' % (k[:50], v['hbm_read_bytes']/1e9, v['hbm_write_bytes']/1e9))"
done
