# the corpus bench line + phase profile per file: bash tools/exp_corpus_files.sh [codec]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
C=${1:-lz4}
for F in $(python -c "
import sys; sys.argv=['x']; sys.path.insert(0,'.'); import bench
cc, files = bench.corpus_chunks(65536); print(' '.join(files))"); do
  echo -n "$F  "
  CJ_FUSED=0 CJ_CORPUS_FILES=$F python bench.py --data corpus64k --codec $C --chunks 40000 --no-cpu-baseline --traffic off --steps 5 --phase-profile 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GB/s %.1f' % d['value'], end='  ')"
  grep -i "cycles/chunk" /tmp/err.txt | cut -c26-95
done
