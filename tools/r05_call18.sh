#!/bin/bash
# Round 5, eighteenth GPU call: the output mask against the commit before it, same box, alternating (512 rays, pooled and plain).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c18; O=gpurun_out/c18
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
cp megastep_amd/csrc/libmegastep_hip.so megastep_amd/csrc/variants/product.so
for rep in 1 2; do for v in prev_nomask product; do
  echo "== $v"; MEGASTEP_HIP_LIB=$PWD/megastep_amd/csrc/variants/$v.so timeout 600 python tools/ab_envstep.py --res 512 --fov 70 --sub 4 --centre 2>> $O/err.txt | grep -E "planes     |obs        |depth"
done; done | tee $O/ab_mask.txt
