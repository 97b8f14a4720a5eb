#!/bin/bash
# Build a library variant whose DEVICE code went through a sed script between compiler and assembler:
#   tools/build_asm_variant.sh NAME 'sed -E script' file.hip [file.hip ...]   -> cramjam_amd/variants/libcramjam_hip_NAME.so
# (the host halves are compiled as usual; the other objects are the product's).  Round 5 used it to turn every
# `v_cndmask_b32_e32 ..., vcc` into the e64 encoding (tools/issue_rate_probe.hip: 16-20 cycles against 4.2 per wavefront instruction).
set -e
NAME=$1; SCRIPT=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd)
L=/opt/rocm/lib/llvm/bin
D=$R/cramjam_amd/variants/obj_$NAME; mkdir -p $D
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function"
OBJS=""
for f in $R/cramjam_amd/build/*.o; do
  b=$(basename $f .o)
  if echo " $* " | grep -q " $b.hip "; then
    /opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only -o $D/$b.s $R/cramjam_amd/csrc/$b.hip 2>/dev/null
    sed -E "$SCRIPT" $D/$b.s > $D/${b}_x.s
    $L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $D/${b}_x.s -o $D/${b}_dev.o
    $L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $D/$b.out $D/${b}_dev.o
    $L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$D/$b.out -output=$D/$b.hipfb
    /opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $D/$b.hipfb -c $R/cramjam_amd/csrc/$b.hip -o $D/$b.o 2>/dev/null
    OBJS="$OBJS $D/$b.o"
  else
    OBJS="$OBJS $f"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/cramjam_amd/variants/libcramjam_hip_$NAME.so $OBJS
echo built $R/cramjam_amd/variants/libcramjam_hip_$NAME.so
