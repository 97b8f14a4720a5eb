# one-off A/B harness (kept for the record): parse kernel time with / without the async line prefetch
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for cfg in "-DCJ_PARSE_ASYNC" "-DCJ_NONE"; do
  CJ_EXTRA_HIPCC_FLAGS="$cfg" python -c "
from cramjam_amd import _build; _build.build(force=True)" 2>&1 | tail -3
  echo "CFG $cfg"
  timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
  rm -rf gpurun_out/ab; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab -- python bench.py --no-cpu-baseline --steps 5 > /dev/null 2>&1
  grep -h "parse_kernel\|lds2" $(find gpurun_out/ab -name "*kernel_stats.csv") | cut -d, -f1-4 | cut -c1-60,150-
  timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-170
done
