# per file of the benchmark corpus: decode rate and the two kernels' times (20 000 chunks of one file's 64 KiB chunks, tiled)
cd $GRAFT_REPO_ROOT
for F in $(python -c "
import sys; sys.argv=['x']; sys.path.insert(0,'.'); import bench
cc, files = bench.corpus_chunks(65536); print(' '.join(files))"); do
  CJ_CORPUS_FILES=$F python bench.py --data corpus64k --codec ${1:-lz4} --chunks 20000 --no-cpu-baseline --traffic off --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline'].get('kernels_ms') or {}
print('%-28s %7.1f GB/s  %7.3f ms  ratio %s  %s' % ('$F', d['value'], d['ms_per_step'], d['config']['ratio'], ' '.join('%s=%.2f' % (n.split('::')[-1][:22], v['ms']) for n, v in k.items())))"
done
