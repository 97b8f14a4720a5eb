#!/bin/bash
# registers / LDS / spills of every kernel of one source file (compiler remarks, no GPU needed)
#   tools/kernel_resources.sh cramjam_amd/csrc/lz4_decode_lds.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Rpass-analysis=kernel-resource-usage "$@" -c "$f" -o /dev/null 2>&1 \
  | grep -E "Function Name|VGPRs:|AGPRs|Spill|LDS Size|Occupancy|SGPRs:" | sed 's/.*remark: [^ ]* //' | paste - - - - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | tr -s ' \t' ' '
