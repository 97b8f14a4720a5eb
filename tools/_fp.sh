cd $GRAFT_REPO_ROOT
echo "new:"; python tests/perf/encoder_fingerprint.py
bash tools/exp_encoders.sh extskip product
