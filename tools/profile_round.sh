cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/v7
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py > gpurun_out/v7/bench.json 2> gpurun_out/v7/bench.err; tail -c 600 gpurun_out/v7/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v7/stats -- python bench.py --no-cpu-baseline > gpurun_out/v7/stats.log 2>&1
find gpurun_out/v7/stats -name "*kernel_stats.csv" | head -2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/v7/fetch -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/v7/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/v7/write -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/v7/write.log 2>&1
F=$(find gpurun_out/v7/fetch -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/v7/write -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $F $W gpurun_out/v7/hbm_traffic.json; cat gpurun_out/v7/hbm_traffic.json | head -40
python bench.py --codec snappy --no-cpu-baseline 2>&1 | tail -1 | cut -c1-220
CJ_SLICE_CHUNKS=131072 python bench.py --chunks 1000000 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-220
