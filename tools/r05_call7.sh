#!/bin/bash
# Round 5, seventh GPU call: vector / scalar / LDS instructions per render wave section by section (the ablation builds under one
# SQ counter pass each), headline shape - is the walk over the list scalar-heavy?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c7; O=gpurun_out/c7
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
cp megastep_amd/csrc/libmegastep_hip.so megastep_amd/csrc/variants/product.so
for v in ab4 ab1 ab2 ab3 nodyn product; do
  MEGASTEP_HIP_LIB=$PWD/megastep_amd/csrc/variants/$v.so bash tools/sq_quick.sh c7_$v 2>&1 | tail -4
done | tee $O/sq_sections.txt
