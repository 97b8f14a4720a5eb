# f04 / f05 on the product build: all GPU tests, then the batch fuzz (mutated chunks through every mapping, parse placement and window)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
echo "product: $(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -1)"
for S in 11 12; do BATCH=1 CASES=150000 SEED=$S timeout 900 python tests/perf/fuzz_large.py 2>&1 | tail -2; done
for M in 32768 16384; do echo "MAXCHUNK=$M"; MAXCHUNK=$M BATCH=1 CASES=150000 SEED=13 timeout 900 python tests/perf/fuzz_large.py 2>&1 | tail -2; done
