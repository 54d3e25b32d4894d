#!/bin/bash
# Round 5, twenty-third GPU call: the wide waves' list capacity (MS_VCAP_WIDE; 110 lines since round 4, chosen with three chunks in flight)
# once more with one chunk in flight: 80 / 96 / 104 / 110 lines at 512 rays and on C5's share.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c23; O=gpurun_out/c23
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
cp megastep_amd/csrc/libmegastep_hip.so megastep_amd/csrc/variants/product.so
bash tools/ab_libs.sh "variants/product variants/vw80 variants/vw96 variants/vw104 variants/product variants/vw96 variants/vw104" "--res 512 --fov 70" "--envs 32768 --agents 1 --res 256 --large --unique 64 --fast-build" 2>&1 | tee $O/ab_vcap_wide.txt
