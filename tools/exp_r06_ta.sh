# round 6: is the encoder bound by its scattered loads?  variants that drop the second forward block / also the backward block of measure's
# first round trip (WRONG streams: --experiment-no-verify), against the product
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for V in base noblk1 noback; do
  L=$PWD/cramjam_amd/variants/libcramjam_hip_$V.so; [ "$V" = base ] && L=$PWD/cramjam_amd/libcramjam_hip.so
  for A in "" "--data corpus64k"; do
  echo -n "$V $A:  "
  CJ_HIP_LIB=$L timeout 600 python bench.py --op compress --no-cpu-baseline --traffic off --steps 8 --experiment-no-verify $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GB/s %.1f  ms/step %.3f ratio %s' % (d['value'], d['ms_per_step'], d['config'].get('ratio')))"
  done
done
