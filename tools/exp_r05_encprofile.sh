#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in w1prof w2prof; do for b in 1 4 9; do
  echo "== $v"; CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so CJ_ENC_BLOCKS=$b python tools/exp_r05_encprofile.py 2>&1 | tail -2
done; done
