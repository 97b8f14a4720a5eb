cd $GRAFT_REPO_ROOT
bash tools/exp_r05_sweep.sh r05_e15 "w1:256 w2eu4:512 product:512" "9:0"
CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_w1prof.so CJ_ENC_BLOCKS=1 python tools/exp_r05_encprofile.py 2>&1 | tail -2
bash tools/exp_r05_occupancy.sh "w1 w2eu4" "1 4 8 9"
