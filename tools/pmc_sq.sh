# SQ counter passes (issue / wait / instruction mix) for one bench.py configuration: bash tools/pmc_sq.sh [bench args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_sq
cd $R
rm -rf gpurun_out/pmc_sq/a gpurun_out/pmc_sq/b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_sq/a -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmc_sq/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d gpurun_out/pmc_sq/b -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmc_sq/b.log 2>&1
python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/pmc_sq/*/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'cj::' in k and 'bench' not in k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in agg:
        print(k)
        for c,v in agg[k].items(): print('   %-24s %.4g (n=%d)'%(c,sum(v)/len(v),len(v)))
PY
