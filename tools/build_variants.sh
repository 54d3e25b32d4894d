#!/bin/bash
# Builds one libmegastep_hip variant per A/B knob setting into megastep_amd/csrc/variants/ (not tracked).
cd "$(dirname "$0")/../megastep_amd/csrc"
mkdir -p variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize"
build() { /opt/rocm/bin/hipcc $FLAGS $2 -o variants/$1.so megastep_hip.hip & }
build v1 "-DMS_V1_OPTS=0"
build v1_noclip "-DMS_V1_OPTS=1"
build v1_flag "-DMS_V1_OPTS=2"
build v1_noclip_flag "-DMS_V1_OPTS=3"
build v2_eager "-DMS_V2_OPTS=1"
build v2_clip "-DMS_V2_OPTS=2"
build v2_noslp "-fno-slp-vectorize"
wait
ls -la variants
