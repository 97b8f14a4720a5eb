# f04: sub-phase cycles of the in-kernel parse (sub-marks 6..10 = P1a, P1b, P2, P3 + scan, P4; 11 = chunks on the list path, per mille)
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-flist0 flist}; do
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  for args in "--chunks 8192 --unique 2048" "--codec snappy --chunks 8192 --unique 2048" "--data corpus64k --chunks 8192"; do
    echo "$v [$args]: $(python bench.py $args --no-cpu-baseline --traffic off --steps 10 --phase-profile 2>&1 | grep 'LDS decoder cycles' | tail -1)"
  done
done
