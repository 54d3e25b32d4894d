#!/bin/bash
# A/B of the render kernel variants (MEGASTEP_RENDER_IMPL) under rocprofv3 --kernel-trace --stats, at the headline,
# C3 and C5-per-GPU shapes. usage: tools/ab_impl.sh "pairs v2" [lib.so ...]
impls=${1:-"pairs v2"}; shift
libs=${@:-megastep_amd/csrc/libmegastep_hip.so}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
lean="--no-cpu-baseline --no-env-fps --no-graph --steps 40 --warmup 5"
for lib in $libs; do for impl in $impls; do
  for shape in "" "--res 128" "--envs 32768 --agents 1 --res 256 --large --unique 64 --fast-build"; do
    MEGASTEP_HIP_LIB=$PWD/$lib MEGASTEP_RENDER_IMPL=$impl rocprofv3 --kernel-trace --stats -d gpurun_out/ab -o r --output-format csv -- python bench.py $lean $shape > gpurun_out/ab.log 2>&1
    echo "== $lib impl=$impl shape='$shape'"
    grep -E "render_kernel|physics_kernel" gpurun_out/ab/r_kernel_stats.csv | awk -F, '{print $1, $2, $4, $6, $7}' | cut -c1-60,150-
    grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab.log | head -1
  done
done; done
