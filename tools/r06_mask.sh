#!/bin/bash
# Round 6: the output mask for the colour instantiation with optional outputs (MS_OBS_MASK) - the Deathmatch shape's render
# (512 rays, pooled RGB-D + crosshair ids: render_kernel<2,1,1,1,4>) with the mask and with the pointers asked, same box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6d; O=gpurun_out/r6d
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
bash tools/build_variants.sh "nomask:-DMS_OBS_MASK=0" > $O/variants.log 2>&1
for rep in 1 2; do for lib in libmegastep_hip variants/nomask; do
  echo "== $lib (pass $rep)"
  MEGASTEP_HIP_LIB=$PWD/megastep_amd/csrc/$lib.so python tools/ab_envstep.py --res 512 --fov 70 --sub 4 --centre 2>&1 | grep -E "plain     render (planes|obs  |obs\+planes)|move\+imu  render obs"
  MEGASTEP_HIP_LIB=$PWD/megastep_amd/csrc/$lib.so python tools/ab_envstep.py --res 64 --fov 130 --sub 1 2>&1 | grep -E "plain     render (planes|obs  )|move\+imu  render obs"
done; done 2>&1 | tee $O/ab_mask.txt
timeout 600 python -m pytest tests/test_gpu_envs.py tests/test_gpu_parity.py tests/test_gpu_groups.py tests/test_gpu_depth.py -q --tb=short -p no:cacheprovider -x > $O/test.log 2>&1; echo "pytest rc=$?"; tail -5 $O/test.log
