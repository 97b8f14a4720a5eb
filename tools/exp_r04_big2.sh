cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r04e; mkdir -p $O
for V in "$@"; do
L=$PWD/cramjam_amd/variants/libcramjam_hip_$V.so; [ "$V" = base ] && L=$PWD/cramjam_amd/libcramjam_hip.so
for C in lz4 snappy; do
CJ_HIP_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$V$C -- python bench.py --codec $C --chunk-bytes 262144 --chunks 8192 --unique 2048 --no-cpu-baseline --traffic off --steps 5 > $O/stats.log 2>&1
python - $V$C <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/r04e/st_%s/**/*kernel_stats.csv' % sys.argv[1], recursive=True)[0]
print(sys.argv[1], ' '.join('%s %.3f' % (r['Name'].split('(')[0].replace('void cj::','')[:24], float(r['AverageNs'])/1e6) for r in csv.DictReader(open(f)) if 'big_walk' in r['Name'] or 'bigslabs' in r['Name']))
PY
done; done
rm -rf $O/st_*
