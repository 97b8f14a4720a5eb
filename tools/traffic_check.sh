cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/v7b
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/v7b/fetch -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/v7b/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/v7b/write -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/v7b/write.log 2>&1
F=$(find gpurun_out/v7b/fetch -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/v7b/write -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $F $W gpurun_out/v7b/hbm_traffic.json > /dev/null; python -c "
import json; d=json.load(open('gpurun_out/v7b/hbm_traffic.json')); print({k:(round(v['hbm_read_bytes']/1e9,2),round(v['hbm_write_bytes']/1e9,2)) for k,v in d['kernels'].items()}, d['total_hbm_bytes_per_step']/1e9)"
