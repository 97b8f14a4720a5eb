# f07: a workgroup's first chunk is its own index (no atomic in front of it)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
echo "tests: $(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -1)"
for args in "--chunk-bytes 32768 --chunks 4096 --unique 2048" "--chunk-bytes 16384 --chunks 4096 --unique 2048" "--chunk-bytes 16384 --chunks 2048 --unique 2048" "--chunk-bytes 16384 --chunks 1024 --unique 1024" "--chunk-bytes 8192 --chunks 4096 --unique 2048" "--chunks 1024 --unique 1024" "--chunks 8192 --unique 2048" "--codec snappy --chunks 8192 --unique 2048" "--codec snappy --chunk-bytes 16384 --chunks 4096 --unique 2048" "--chunk-bytes 16384 --chunks 16384 --unique 2048" "--chunk-bytes 32768 --chunks 16384 --unique 2048" "--chunk-bytes 16384 --chunks 400000" ""; do
  python bench.py $args --no-cpu-baseline --traffic off --steps 30 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$args]: %.1f GB/s %.3f ms' % (d['value'], d['ms_per_step']))"
done
for args in "--chunk-bytes 16384 --chunks 4096 --unique 2048" "--chunks 4096 --unique 2048"; do
  echo "[$args]: $(python bench.py $args --no-cpu-baseline --traffic off --steps 10 --phase-profile 2>&1 | grep 'LDS decoder cycles' | tail -1)"
done
