# one-off A/B harness (kept for the record)
cd $GRAFT_REPO_ROOT
for cfg in "-DCJ_D3_SLEEP=1" "-DCJ_D3_SLEEP=2" "-DCJ_D3_SLEEP=4" "-DCJ_NONE"; do
  CJ_EXTRA_HIPCC_FLAGS="$cfg" python -c "
from cramjam_amd import _build; _build.build(force=True)" 2>&1 | tail -3
  echo "CFG $cfg"
  timeout 300 python bench.py --lz4-mode lds --phase-profile --no-cpu-baseline 2>&1 | tail -2 | cut -c1-170
done
