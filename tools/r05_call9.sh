#!/bin/bash
# Round 5, ninth GPU call: one chunk of rows in flight (MS_AHEAD=1) with six and with seven waves a SIMD (68 VGPRs and a list of 85
# lines = 5112 B of LDS = four 1280 B granules let seven in; round 3's "seven waves" still had five granules - never seven waves).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c9; O=gpurun_out/c9
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
cp megastep_amd/csrc/libmegastep_hip.so megastep_amd/csrc/variants/product.so
bash tools/ab_libs.sh "variants/product variants/a1 variants/a1w7 variants/a1w7v64 variants/a3w6v85 variants/product variants/a1 variants/a1w7 variants/a1w7v64" "" "--agents 1" "--agents 1 --depth-only" "--res 128 --fov 70" 2>&1 | tee $O/ab_waves.txt
