# x04: lane_copy with all size classes' loads before any store (a flush was four dependent global round trips); `base` = the variant built from the commit before
cd $GRAFT_REPO_ROOT
for V in base new; do
[ $V = base ] && export CJ_HIP_LIB=$PWD/cramjam_amd/variants/libcramjam_hip_base.so || unset CJ_HIP_LIB
for C in lz4 snappy; do
python bench.py --op compress --codec $C --no-cpu-baseline --traffic off --steps 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$V synth $C %.1f GB/s' % d['value'])"
python bench.py --op compress --data corpus64k --codec $C --chunks 20000 --no-cpu-baseline --traffic off --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$V corpus $C %.1f GB/s' % d['value'])"
done
done
