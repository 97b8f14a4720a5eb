# TA / TCP counter passes (texture addresser = the vector memory front end of a CU) for one bench.py configuration:
#   bash tools/pmc_ta.sh [bench args...]      (CJ_HIP_LIB selects a variant).  At most two counters of a block per pass (more: rocprofv3
#   aborts with "exceeds the capabilities of the hardware" and then hangs — every pass under its own timeout).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_ta
cd $R
rm -rf gpurun_out/pmc_ta/p*
i=0
for C in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_ta/p$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --traffic off "$@" > gpurun_out/pmc_ta/p$i.log 2>&1 || echo "pass $i ($C) failed: $(grep -m1 'error code' gpurun_out/pmc_ta/p$i.log)"
done
python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/pmc_ta/p*/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'cj::' in k and 'bench' not in k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in agg:
        print(k[:70])
        for c,v in agg[k].items(): print('   %-40s %.5g (n=%d)'%(c,sum(v)/len(v),len(v)))
PY
