#!/bin/bash
# Round 5, nineteenth GPU call: the output mask for the colourless instantiations only (through obs_subsample's upper half): suite (also
# under the older raycasts), fuzz, A/B against the commit before the mask on one box.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c19; O=gpurun_out/c19
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/test.log 2>&1; echo "pytest rc=$?"; tail -3 $O/test.log
make -C megastep_amd/csrc ab > $O/make_ab.log 2>&1
for impl in pairs seq; do
  MEGASTEP_HIP_LIB=$PWD/megastep_amd/csrc/libmegastep_hip_ab.so MEGASTEP_RENDER_IMPL=$impl timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "not scale and not full_benchmark" > $O/test_$impl.log 2>&1; echo "pytest under $impl rc=$?"; tail -1 $O/test_$impl.log
done
timeout 900 python tools/fuzz_parity.py 15000 400 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
cp megastep_amd/csrc/libmegastep_hip.so megastep_amd/csrc/variants/product.so
for rep in 1 2; do for v in prev_nomask product; do
  echo "== $v"; MEGASTEP_HIP_LIB=$PWD/megastep_amd/csrc/variants/$v.so timeout 600 python tools/ab_envstep.py --res 512 --fov 70 --sub 4 --centre 2>> $O/err.txt | grep -E "plain     render (planes|obs|depth)"
done; done | tee $O/ab_mask.txt
for v in prev_nomask product prev_nomask product; do echo "== $v C2"; MEGASTEP_HIP_LIB=$PWD/megastep_amd/csrc/variants/$v.so timeout 600 python tools/ab_envstep.py --agents 1 2>> $O/err.txt | grep -E "plain     render (planes|obs|depth)"; done | tee $O/ab_mask_c2.txt
