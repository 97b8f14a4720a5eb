#!/bin/bash
# issue / wait counters of the encoder kernels (per chunk): bash tools/pmc_encode.sh TAG [variant ...]
# (variant "product" = in-tree library; others = cramjam_amd/variants/libcramjam_hip_<variant>.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
[ $# -eq 0 ] && set -- product
SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES"
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAVES"
     )   # TA_* / TCP_* sets: the profiler never returns with them on this pool (measured: 800 s, killed)
for v in "$@"; do
  export CJ_HIP_LIB=$R/cramjam_amd/variants/libcramjam_hip_$v.so
  [ "$v" = "product" ] && unset CJ_HIP_LIB
  for codec in lz4 snappy; do
    for i in 0 1; do
      d=$O/pmc_${v}_${codec}_$i
      rm -rf $d
      rocprofv3 --pmc ${SETS[$i]} --output-format csv -d $d -- python bench.py --op compress --codec $codec --chunks 20000 --steps 2 --warmup 1 --no-cpu-baseline --traffic off > $d.log 2>&1
      python - "$d" "$v" <<'PY' | tee -a $O/pmc_encode_summary.txt
import csv,glob,collections,sys
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'cj::' in k and 'encode' in k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
tot=collections.defaultdict(float)
for k in agg:
    for c,v in agg[k].items(): tot[c]+=sum(v)/len(v)          # per launch; a step = the LDS-table kernel + the global-table kernel
    print(sys.argv[2], k, ' '.join('%s=%.0f' % (c.replace('SQ_',''), sum(v)/len(v)/2e4) for c,v in sorted(agg[k].items())), '(per launch / 20 000 chunks)')
print(sys.argv[2], 'all encoder kernels', ' '.join('%s=%.0f' % (c.replace('SQ_',''), v/2e4) for c,v in sorted(tot.items())), '(per chunk)')
PY
    done
  done
done
