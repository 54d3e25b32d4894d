#!/bin/bash
# Builds libmegastep_hip variants into megastep_amd/csrc/variants/ (not tracked) and prints the render kernel's resources.
# usage: tools/build_variants.sh "name:-DFLAG=.. -DFLAG=.." ...
cd "$(dirname "$0")/../megastep_amd/csrc"; mkdir -p variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize"
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  ( /opt/rocm/bin/hipcc $F $flags -Rpass-analysis=kernel-resource-usage -o variants/$name.so megastep_hip.hip > /tmp/bv_$name.log 2>&1
    grep -E "error" /tmp/bv_$name.log | head -3
    grep -E "Function Name|VGPRs:|Spill|ScratchSize|Occupancy" /tmp/bv_$name.log | sed 's/.*remark: //; s/ \[-Rpass.*//' | paste - - - - - - | grep "render_kernelILi2ELi1ELi0" | sed "s/^.*RenderConstsE/$name:/" | tr -s ' \t' ' ' ) &
done; wait
