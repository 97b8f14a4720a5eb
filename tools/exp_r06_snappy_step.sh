# f06: the Snappy straight-line step with literal headers of up to 4 bytes and copy-4 elements (parse_grammar.hpp: walk_step / walk_step_carry)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
echo "tests: $(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -1)"
for args in "--codec snappy --chunks 1024 --unique 1024" "--codec snappy --chunks 8192 --unique 2048" "--codec snappy --chunks 16384 --unique 2048" "--data corpus64k --codec snappy --chunks 8192" "--codec snappy --chunk-bytes 32768 --chunks 4096 --unique 2048" "--codec snappy --chunk-bytes 16384 --chunks 4096 --unique 2048" "--chunks 8192 --unique 2048"; do
  python bench.py $args --no-cpu-baseline --traffic off --steps 30 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$args]: %.1f GB/s %.3f ms' % (d['value'], d['ms_per_step']))"
done
for args in "--codec snappy --chunks 8192 --unique 2048" "--data corpus64k --codec snappy --chunks 8192"; do
  echo "[$args]: $(python bench.py $args --no-cpu-baseline --traffic off --steps 10 --phase-profile 2>&1 | grep 'LDS decoder cycles' | tail -1)"
done
python tests/perf/single_buffer_rates.py 2>&1 | tail -12
BATCH=1 CASES=150000 SEED=21 timeout 900 python tests/perf/fuzz_large.py 2>&1 | tail -2
CASES=8000 SEED=22 timeout 900 python tests/perf/fuzz_large.py 2>&1 | tail -2
