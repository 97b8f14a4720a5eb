# SQ counters of the decode kernels for one environment setting:  bash tools/exp_r04_pmc.sh "CJ_SEG_KLOG=1" [bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
E=$1; shift
O=gpurun_out/pmc_tmp; rm -rf $O; mkdir -p $O
env $E rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $O -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --traffic off "$@" > $O/log.txt 2>&1
python - "$O" <<'PY'
import csv,glob,collections,sys
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'cj::' in k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print(k[:60], ' '.join('%s=%.3g' % (c.replace('SQ_',''), sum(v)/len(v)) for c,v in sorted(agg[k].items())))
PY
rm -rf $O
