# one-off A/B harness used while tuning the LDS decoder (kept for the record)
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --lz4-mode lds --phase-profile --no-cpu-baseline 2>&1 | tail -2 | cut -c1-170
