#!/bin/bash
# Build a tuning variant of the library next to the product one: tools/build_variant.sh NAME "-DFLAG=1 ..." [sources...]
# -> cramjam_amd/variants/libcramjam_hip_NAME.so (only the listed sources are recompiled with the flags; default: lz4_decode_lds.hip).
# Select it at run time with CJ_HIP_LIB=<path> (cramjam_amd/_native.py).  Dev tool: variants are never committed.
set -e
NAME=$1; FLAGS=$2; shift 2 || true
SRCS=${@:-lz4_decode_lds.hip}
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/cramjam_amd/variants/obj_$NAME
OBJS=""
for f in $R/cramjam_amd/build/*.o; do
  b=$(basename $f .o)
  if echo " $SRCS " | grep -q " $b.hip "; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function $FLAGS -c $R/cramjam_amd/csrc/$b.hip -o $R/cramjam_amd/variants/obj_$NAME/$b.o
    OBJS="$OBJS $R/cramjam_amd/variants/obj_$NAME/$b.o"
  else
    OBJS="$OBJS $f"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/cramjam_amd/variants/libcramjam_hip_$NAME.so $OBJS
echo built $R/cramjam_amd/variants/libcramjam_hip_$NAME.so
