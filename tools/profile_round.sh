# Round-end evidence run (on the GPU box): gpu tests, default bench line, rocprofv3 kernel stats of the same command,
# FETCH_SIZE / WRITE_SIZE passes (separate, as the MI355X guide prescribes) and the secondary-path bench lines.
# Outputs land in gpurun_out/<tag>/; copy the summaries into profiles/rNN/.
TAG=${1:-v7}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 700 $O/bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline > $O/stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; head -8 $O/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/write.log 2>&1
cp $(find $O/fetch -name "*counter_collection.csv" | head -1) $O/fetch_size_counter_collection.csv
cp $(find $O/write -name "*counter_collection.csv" | head -1) $O/write_size_counter_collection.csv
python tools/pmc_summary.py $O/fetch_size_counter_collection.csv $O/write_size_counter_collection.csv $O/hbm_traffic.json > /dev/null
python -c "
import json; d=json.load(open('$O/hbm_traffic.json')); print({k:(round(v['hbm_read_bytes']/1e9,2),round(v['hbm_write_bytes']/1e9,2)) for k,v in d['kernels'].items()}, d['total_hbm_bytes_per_step']/1e9)"
for args in "--codec snappy" "--op compress" "--op compress --codec snappy" "--chunks 1000000"; do
  python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 >> $O/other_paths.jsonl
done
cut -c1-175 $O/other_paths.jsonl
rm -rf $O/stats $O/fetch $O/write
