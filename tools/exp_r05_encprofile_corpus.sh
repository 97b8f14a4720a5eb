cd $GRAFT_REPO_ROOT
export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_w2prof.so
for f in "" mr kppkn.gtb alice29.txt geo.protodata xml; do CORPUS_FILE=$f CHUNKS=4000 python tools/exp_r05_encprofile.py 2>&1 | tail -2; done
