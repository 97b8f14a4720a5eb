cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for v in "" XD3 XD23; do
  L=""; [ -n "$v" ] && L=$R/cramjam_amd/variants/libcramjam_hip_$v.so
  rm -rf gpurun_out/pmc_x; mkdir -p gpurun_out/pmc_x
  CJ_HIP_LIB=$L rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_BRANCH --kernel-trace --output-format csv -d gpurun_out/pmc_x -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --traffic off --experiment-no-verify > gpurun_out/pmc_x.log 2>&1
  echo "== variant $v"
  python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_x/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'lds2_kernel' in k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in agg:
    print(k, ' '.join('%s=%.0f' % (c.replace('SQ_',''), sum(v)/len(v)/1e5) for c,v in sorted(agg[k].items())))
PY
  CJ_HIP_LIB=$L python bench.py --steps 5 --warmup 1 --no-cpu-baseline --traffic off --experiment-no-verify --phase-profile 2>&1 | grep -E "cycles/chunk|ms_per_step" | cut -c1-200 | sed "s/.*\"value\": \([0-9.]*\).*ms_per_step\": \([0-9.]*\).*/value \1 ms_per_step \2/"
done
