# phase cycles of the one-kernel path on small windows
cd $GRAFT_REPO_ROOT
for args in "--chunk-bytes 32768 --chunks 4096 --unique 2048" "--chunk-bytes 16384 --chunks 4096 --unique 2048" "--chunk-bytes 16384 --chunks 2048 --unique 2048" "--chunk-bytes 8192 --chunks 4096 --unique 2048" "--chunks 4096 --unique 2048"; do
  echo "[$args]: $(python bench.py $args --no-cpu-baseline --traffic off --steps 10 --phase-profile 2>&1 | grep 'LDS decoder cycles' | tail -1)"
done
