cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r04b; mkdir -p $O
for K in 0 1 2 3; do
  echo "== klog $K"; CJ_SEG_KLOG=$K python bench.py --no-cpu-baseline --traffic off --steps 20 2>$O/err_$K.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
echo "== old"; CJ_PARSE=old python bench.py --no-cpu-baseline --traffic off --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
for K in 1 2; do
CJ_SEG_KLOG=$K rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$K -- python bench.py --no-cpu-baseline --traffic off --steps 10 > $O/stats$K.log 2>&1
cp $(find $O/stats$K -name "*kernel_stats.csv" | head -1) $O/kernel_stats_k$K.csv; head -4 $O/kernel_stats_k$K.csv | cut -c1-60,200-330
done
CJ_SEG_KLOG=2 python bench.py --no-cpu-baseline --traffic off --steps 10 --phase-profile 2>&1 | grep -i "cycles/chunk"
rm -rf $O/stats1 $O/stats2
