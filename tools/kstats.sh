#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
MEGASTEP_HIP_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats -d gpurun_out/ks -o r --output-format csv -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline $BENCH_ARGS > /dev/null 2>&1
echo "== $lib"; grep -E "render_kernel|dynlight|physics" gpurun_out/ks/r_kernel_stats.csv | cut -c1-175
done
