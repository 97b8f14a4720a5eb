# Round-end evidence run (on the GPU box): gpu tests, the default bench line (it measures its HBM traffic itself with two
# rocprofv3 --pmc child runs), rocprofv3 kernel stats of the same command, SQ / LDS counters, and the secondary-path bench lines.
# Outputs land in gpurun_out/<tag>/; copy the summaries into profiles/rNN/.
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3 | tee $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline --traffic off > $O/stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; head -8 $O/kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_enc -- python bench.py --op compress --no-cpu-baseline --traffic off --steps 10 > $O/stats_enc.log 2>&1
cp $(find $O/stats_enc -name "*kernel_stats.csv" | head -1) $O/kernel_stats_compress.csv; head -5 $O/kernel_stats_compress.csv
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --traffic off > $O/sq.log 2>&1
# unit-busy counters of the same kernels in their own pass (quad-cycles a SIMD / the scalar unit / the LDS spent executing)
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD --output-format csv -d $O/sq2 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --traffic off > $O/sq2.log 2>&1
for c in lz4 snappy; do rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq3_$c -- python bench.py --op compress --codec $c --steps 2 --warmup 1 --no-cpu-baseline --traffic off > $O/sq3_$c.log 2>&1; done
python - "$O/sq" "$O/sq2" "$O/sq3_lz4" "$O/sq3_snappy" <<'PY' | tee $O/sq_counters_per_chunk.txt
import csv,glob,collections,sys,os
for d in sys.argv[1:]:
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0]
            want = ('encode' in k) if 'sq3' in d else ('encode' not in k)          # the compress passes run the decoders only to verify
            if 'cj::' in k and want: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    print('#', os.path.basename(d), '(default decompress batch)' if 'sq3' not in d else '(compress batch)')
    for k in sorted(agg):
        print(k, ' '.join('%s=%.0f' % (c.replace('SQ_',''), max(v)/1e5) for c,v in sorted(agg[k].items())), '(per chunk of the 100 k; the launch over the full batch)')
PY
# the corpus lines carry their own traffic and cpu_baseline (all 20 files, liblz4 / libsnappy streams)
python bench.py --data corpus64k --steps 20 --traffic on --cpu-seconds 10 2>/dev/null | tail -1 >> $O/other_paths.jsonl
python bench.py --data corpus64k --codec snappy --steps 20 --traffic on --cpu-seconds 10 2>/dev/null | tail -1 >> $O/other_paths.jsonl
for args in "--codec snappy" "--op compress" "--op compress --codec snappy" "--codec snappy --op roundtrip" "--chunks 1000000 --steps 20" "--workload mixed256k --steps 20" "--codec lz4 --chunk-bytes 262144 --chunks 8192 --unique 2048 --steps 20" "--codec snappy --chunk-bytes 262144 --chunks 8192 --unique 2048 --steps 20" "--chunks 16384 --unique 2048" "--chunks 8192 --unique 2048" "--chunks 1024 --unique 1024" "--codec snappy --chunks 8192 --unique 2048" "--codec snappy --chunks 1024 --unique 1024" "--chunk-bytes 32768 --chunks 4096 --unique 2048" "--chunk-bytes 16384 --chunks 4096 --unique 2048" "--codec snappy --chunk-bytes 32768 --chunks 4096 --unique 2048" "--codec snappy --chunk-bytes 16384 --chunks 4096 --unique 2048" "--data corpus64k --chunks 8192" "--chunk-bytes 32768 --chunks 200000" "--chunk-bytes 16384 --chunks 400000" "--codec snappy --chunk-bytes 32768 --chunks 200000" "--codec snappy --chunk-bytes 16384 --chunks 400000" "--chunk-bytes 32768 --chunks 16384 --unique 2048" "--chunk-bytes 16384 --chunks 16384 --unique 2048" "--op compress --data corpus64k --steps 10" "--op compress --data corpus64k --codec snappy --steps 10"; do
  python bench.py --cpu-seconds 6 --traffic off $args 2>/dev/null | tail -1 >> $O/other_paths.jsonl
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_mixed -- python bench.py --workload mixed256k --no-cpu-baseline --traffic off --steps 10 > $O/stats_mixed.log 2>&1
cp $(find $O/stats_mixed -name "*kernel_stats.csv" | head -1) $O/kernel_stats_mixed256k.csv; head -8 $O/kernel_stats_mixed256k.csv | cut -c1-70,200-300
python - <<PY
import json
for l in open('$O/other_paths.jsonl'):
    d=json.loads(l); print('%-95s %8.1f GB/s  %8.3f ms/step  frac %.4f' % (d['config']['workload'][:95], d['value'], d['ms_per_step'], d['roofline']['frac']))
PY
timeout 600 python tests/perf/device_api_rate.py 2>&1 | tail -3 > $O/device_api_rate.txt; cat $O/device_api_rate.txt
rm -rf $O/stats $O/sq $O/sq2 $O/sq3_lz4 $O/sq3_snappy $O/stats_enc $O/stats_mixed
