# instruction-cache and wait-breakdown counters for one bench.py configuration: bash tools/pmc_icache.sh [bench args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_ic
cd $R
rm -rf gpurun_out/pmc_ic/a gpurun_out/pmc_ic/b
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_BRANCH SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/pmc_ic/a -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmc_ic/a.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d gpurun_out/pmc_ic/b -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmc_ic/b.log 2>&1
python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/pmc_ic/*/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'cj::' in k and 'bench' not in k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in agg:
        print(k)
        for c,v in agg[k].items(): print('   %-28s %.4g (n=%d)'%(c,sum(v)/len(v),len(v)))
PY
