#!/bin/bash
# Round 5, twelfth GPU call: the older raycasts (make ab) through the suite on the final sources, a long fuzz, two ray groups a wave at 128 rays again
# (with one chunk in flight), smoke.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c12; O=gpurun_out/c12
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build_smoke.log 2>&1; echo "build+smoke rc=$?"; tail -1 $O/build_smoke.log
make -C megastep_amd/csrc ab > $O/make_ab.log 2>&1; echo "make ab rc=$?"
for impl in pairs seq; do
  MEGASTEP_HIP_LIB=$PWD/megastep_amd/csrc/libmegastep_hip_ab.so MEGASTEP_RENDER_IMPL=$impl timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "not scale and not full_benchmark" > $O/test_$impl.log 2>&1; echo "pytest under $impl rc=$?"; tail -1 $O/test_$impl.log
done
timeout 1200 python tools/fuzz_parity.py 10000 1200 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
timeout 600 python tools/ab_groups.py --groups=1,2 c3 2> $O/ab_groups_c3.err | tee $O/ab_groups_c3.txt
