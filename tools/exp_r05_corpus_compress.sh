#!/bin/bash
# round 5: the encoders on the reference's benchmark corpus, per file: bash tools/exp_r05_corpus_compress.sh [codec]
cd $GRAFT_REPO_ROOT
C=${1:-lz4}
for F in $(python -c "
import sys; sys.argv=['x']; sys.path.insert(0,'.'); import bench
cc, files = bench.corpus_chunks(65536); print(' '.join(files))"); do
  CJ_CORPUS_FILES=$F python bench.py --op compress --data corpus64k --codec $C --chunks 20000 --no-cpu-baseline --traffic off --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %7.1f GB/s  ratio %s' % ('$F', d['value'], d['config']['ratio']))"
done
