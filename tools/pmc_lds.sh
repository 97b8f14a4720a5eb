#!/bin/bash
# LDS / issue counters of the decode kernel for tuning variants: bash tools/pmc_lds.sh TAG name1 name2 ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for v in "$@"; do
  export CJ_HIP_LIB=$R/cramjam_amd/variants/libcramjam_hip_$v.so
  [ "$v" = "product" ] && unset CJ_HIP_LIB
  rm -rf $O/pmc_$v
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_$v -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_$v.log 2>&1
  python - "$O/pmc_$v" "$v" <<'PY' | tee -a $O/pmc_summary.txt
import csv,glob,collections,sys
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'lds2' in k or 'parse_kernel' in k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in agg:
    print(sys.argv[2], k, ' '.join('%s=%.0f' % (c.replace('SQ_',''), sum(v)/len(v)/1e5) for c,v in sorted(agg[k].items())), '(per chunk)')
PY
done
