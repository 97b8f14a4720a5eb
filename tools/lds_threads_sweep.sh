cd $GRAFT_REPO_ROOT
for t in 384 640 512; do
  CJ_EXTRA_HIPCC_FLAGS="-DCJ_L2_THREADS=$t" python -c "
from cramjam_amd import _build; _build.build(force=True)" 2>&1 | tail -3
  echo "THREADS $t"
  timeout 300 python bench.py --lz4-mode lds --phase-profile --no-cpu-baseline 2>&1 | tail -2 | cut -c1-170
done
