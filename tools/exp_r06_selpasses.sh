# round 6: parallel selection passes before the serial walk (8 / 12 / 16 / 24), corpus and synth compress
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for V in base sel12 sel16 sel24; do
  L=$PWD/cramjam_amd/variants/libcramjam_hip_$V.so; [ "$V" = base ] && L=$PWD/cramjam_amd/libcramjam_hip.so
  for A in "--data corpus64k" "" "--data corpus64k --codec snappy"; do
  echo -n "$V $A:  "
  CJ_HIP_LIB=$L timeout 600 python bench.py --op compress --no-cpu-baseline --traffic off --steps 8 $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GB/s %.1f  ms/step %.3f ratio %s' % (d['value'], d['ms_per_step'], d['config'].get('ratio')))"
  done
done
