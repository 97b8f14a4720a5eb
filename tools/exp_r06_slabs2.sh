# round 6: per-kernel times of the big-chunk path, 32 KiB slabs (product) against 64 KiB slabs (variant slab64)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for V in base; do
for CODEC in lz4 snappy; do
  L=$PWD/cramjam_amd/variants/libcramjam_hip_$V.so; [ "$V" = base ] && L=$PWD/cramjam_amd/libcramjam_hip.so
  echo "== $V $CODEC"
  rm -rf /tmp/st; CJ_HIP_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python bench.py --chunk-bytes 262144 --chunks 8192 --codec $CODEC --no-cpu-baseline --traffic off --steps 5 > /tmp/st.log 2>&1
  python - <<'PY'
import csv,glob
f=glob.glob('/tmp/st/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'cj::' in r['Name'] and float(r['AverageNs']) > 20000: print('  ', r['Name'][:60], r['Calls'], '%.3f ms' % (float(r['AverageNs'])/1e6))
PY
  CJ_HIP_LIB=$L timeout 600 python bench.py --chunk-bytes 262144 --chunks 8192 --codec $CODEC --no-cpu-baseline --traffic off --steps 5 --phase-profile 2>&1 | grep "cycles/chunk" | head -1 | cut -c1-150
done
done
