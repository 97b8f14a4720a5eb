cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r04d; mkdir -p $O
python bench.py --workload mixed256k --no-cpu-baseline --traffic off --steps 10 2>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mixed256k GB/s %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
tail -3 $O/err.txt
python bench.py --workload mixed256k --lz4-mode wave --no-cpu-baseline --traffic off --steps 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wave-only GB/s %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --workload mixed256k --no-cpu-baseline --traffic off --steps 5 > $O/stats.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r04d/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'cj::' in r['Name']: print(r['Name'][:70], r['Calls'], '%.3f ms' % (float(r['AverageNs'])/1e6))
PY
rm -rf $O/stats
