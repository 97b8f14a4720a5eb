cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r04e; mkdir -p $O
for C in lz4 snappy; do
python bench.py --codec $C --chunk-bytes 262144 --chunks 8192 --unique 2048 --no-cpu-baseline --traffic off --steps 10 2>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$C 8192x256k GB/s %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$C -- python bench.py --codec $C --chunk-bytes 262144 --chunks 8192 --unique 2048 --no-cpu-baseline --traffic off --steps 5 > $O/stats.log 2>&1
python - $C <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/r04e/stats_%s/**/*kernel_stats.csv' % sys.argv[1], recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'cj::' in r['Name']: print('   ', r['Name'][:60], r['Calls'], '%.3f ms' % (float(r['AverageNs'])/1e6))
PY
done
rm -rf $O/stats_*
