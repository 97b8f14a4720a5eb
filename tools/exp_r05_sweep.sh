#!/bin/bash
# round 5: encoder configuration sweep — bash tools/exp_r05_sweep.sh TAG "variant:R ..." "lds:tab ..."
# (variant "product" = in-tree library; R = positions per round of that build, for the byte-for-byte test against the model)
cd $GRAFT_REPO_ROOT
TAG=$1; O=gpurun_out/$TAG; mkdir -p $O
for vr in $2; do
  v=${vr%%:*}; R=${vr##*:}
  export CJ_HIP_LIB=$GRAFT_REPO_ROOT/cramjam_amd/variants/libcramjam_hip_$v.so
  [ "$v" = "product" ] && unset CJ_HIP_LIB
  CJ_TEST_ENC2_R=$R timeout 600 python -m pytest tests/test_enc2_gpu.py -x -q 2>&1 | tail -3 | sed "s/^/$v: /" | tee -a $O/summary.txt
  for lt in $3; do
    export CJ_ENC_LDS_BLOCKS=${lt%%:*} CJ_ENC_TABLE_BLOCKS=${lt##*:}
    for codec in lz4 snappy; do
      timeout 300 python bench.py --op compress --codec $codec --no-cpu-baseline --traffic off --steps 5 --warmup 2 > $O/b.json 2> $O/b.err
      python -c "
import json,sys
t=open('$O/b.json').read().strip()
if not t: print('$v $lt $codec: no JSON', open('$O/b.err').read().strip().splitlines()[-1][:300]); sys.exit(0)
d=json.loads(t.splitlines()[-1]); print('$v lds:tab=$lt $codec %.1f GB/s ms/step %.3f ratio %s' % (d['value'], d['ms_per_step'], d['config'].get('ratio')))" | tee -a $O/summary.txt
    done
  done
  unset CJ_ENC_LDS_BLOCKS CJ_ENC_TABLE_BLOCKS
done
