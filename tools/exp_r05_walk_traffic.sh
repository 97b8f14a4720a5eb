# b-traffic: HBM bytes per kernel of the big-chunk path (8 192 x 256 KiB LZ4): is the walk's 16-byte window re-fetching its lines?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
rm -rf /tmp/pmc_$C
timeout 600 rocprofv3 --pmc $C -d /tmp/pmc_$C -o out --output-format csv -- python $R/bench.py --codec lz4 --chunk-bytes 262144 --chunks 8192 --unique 2048 --no-cpu-baseline --traffic off --steps 2 --warmup 1 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob('/tmp/pmc_$C/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] != '$C': continue
    k = r['Kernel_Name'].split('(')[0][:60]
    acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
for k, (v, n) in sorted(acc.items(), key=lambda x: -x[1][0])[:8]:
    mul = 2.0 if '$C' == 'FETCH_SIZE' else 1.0
    print('$C  %-62s launches %3d  %.3f GB per launch' % (k, n, v * 1024 * mul / n / 1e9))
PY
done
