# A/B harness for compile-time variants: rebuild with CJ_EXTRA_HIPCC_FLAGS and print the LDS decoder's
# per-phase cycle counters.  usage on the GPU box: bash tools/lds_phase_profile.sh "-DSOME_FLAG" "-DCJ_NONE"
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  CJ_EXTRA_HIPCC_FLAGS="$cfg" python -c "
from cramjam_amd import _build; _build.build(force=True)" 2>&1 | tail -3
  echo "CFG $cfg"
  timeout 300 python bench.py --lz4-mode lds --phase-profile --no-cpu-baseline 2>&1 | tail -2 | cut -c1-170
done
