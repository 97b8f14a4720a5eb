# round 6: cycle counters around the matcher's phases (-DCJ_ENC_PROFILE variant encprof), synth and corpus files
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export CJ_HIP_LIB=$PWD/cramjam_amd/variants/libcramjam_hip_encprof.so
python tools/exp_r05_encprofile.py 2>&1 | tail -3
for F in alice29.txt html kppkn.gtb mr urls.10K geo.protodata; do CORPUS_FILE=$F python tools/exp_r05_encprofile.py 2>&1 | tail -3; done
