# per-kernel times of the mixed workload with three groups per codec (49 152 chunks): product against a variant
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${VARIANTS:-product base}; do
  export CJ_HIP_LIB=$R/cramjam_amd/variants/libcramjam_hip_$v.so; [ $v = product ] && unset CJ_HIP_LIB
  rm -rf /tmp/ks_$v
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_$v -o out --output-format csv -- python $R/bench.py --workload mixed256k --chunks ${CHUNKS:-49152} --no-cpu-baseline --traffic off --steps 6 --warmup 2 > /tmp/ks_$v.log 2>&1
  python - <<PY
import csv, glob, json
line = [l for l in open('/tmp/ks_$v.log') if l.startswith('{')]
print("== $v", json.loads(line[-1])['value'] if line else 'no bench line')
f = glob.glob('/tmp/ks_$v/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print('  %-60s calls %4s avg %10.1f us total %9.1f ms' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
done
