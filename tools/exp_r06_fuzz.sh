# round 6: the fuzz campaigns on the round's last product commit — batches of mutated chunks through every mapping and window (oracle's verdict and bytes),
# the encoders against the model, the large-stream and frame decoders
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
( for S in 1 2 3; do BATCH=1 CASES=200000 SEED=$S timeout 900 python tests/perf/fuzz_large.py 2>&1 | tail -2; done
  for M in 32768 16384; do for S in 4 5; do echo "MAXCHUNK=$M"; MAXCHUNK=$M BATCH=1 CASES=150000 SEED=$S timeout 900 python tests/perf/fuzz_large.py 2>&1 | tail -2; done; done ) > $O/r06_fuzz_batch.txt 2>&1
N=20000 timeout 1500 python tests/perf/fuzz_enc2.py > $O/r06_fuzz_enc2.txt 2>&1
( CASES=20000 SEED=7 timeout 900 python tests/perf/fuzz_large.py 2>&1 | tail -2; FRAMES=1 CASES=20000 SEED=8 timeout 900 python tests/perf/fuzz_large.py 2>&1 | tail -2; COMPRESS=1 CASES=20000 SEED=9 timeout 900 python tests/perf/fuzz_large.py 2>&1 | tail -2 ) > $O/r06_fuzz_large.txt 2>&1
tail -3 $O/r06_fuzz_batch.txt $O/r06_fuzz_enc2.txt $O/r06_fuzz_large.txt
