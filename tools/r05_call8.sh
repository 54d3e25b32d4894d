#!/bin/bash
# Round 5, eighth GPU call: chunks of rows in flight in the walk (MS_AHEAD = 1, 2, 3 = the product) per shape.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c8; O=gpurun_out/c8
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
cp megastep_amd/csrc/libmegastep_hip.so megastep_amd/csrc/variants/product.so
bash tools/ab_libs.sh "variants/product variants/ahead1 variants/ahead2 variants/product variants/ahead1 variants/ahead2" "" "--agents 1" "--res 128 --fov 70" "--res 512 --fov 70" "--envs 32768 --agents 1 --res 256 --large --unique 64 --fast-build" 2>&1 | tee $O/ab_ahead.txt
