#!/bin/bash
# Round 5, fifth GPU call: where the headline render launch's TIME goes - the kernel cut off after its block mapping (ab4), after
# its set-up (ab1), after pass 1 (ab2), after the raycast (ab3), without dynamic lighting (nodyn), whole (product).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c5; O=gpurun_out/c5
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
cp megastep_amd/csrc/libmegastep_hip.so megastep_amd/csrc/variants/product.so
bash tools/ab_libs.sh "variants/product variants/ab4 variants/ab1 variants/ab2 variants/ab3 variants/nodyn variants/product" "" "--agents 1 --depth-only" "--res 512 --fov 70" 2>&1 | tee $O/ablate.txt
