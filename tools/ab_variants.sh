#!/bin/bash
# Kernel time of the render kernel per library variant (tools/build_variants.sh) and MEGASTEP_RENDER_IMPL, headline shape,
# under rocprofv3 --kernel-trace --stats. usage: tools/ab_variants.sh "lib:impl lib:impl ..." [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pairs=$1; shift
lean="--no-cpu-baseline --no-env-fps --no-graph --steps 60 --warmup 5 $@"
for pi in $pairs; do
  lib=${pi%%:*}; impl=${pi##*:}
  path=$PWD/megastep_amd/csrc/variants/$lib.so; [ "$lib" = "main" ] && path=$PWD/megastep_amd/csrc/libmegastep_hip.so
  MEGASTEP_HIP_LIB=$path MEGASTEP_RENDER_IMPL=$impl rocprofv3 --kernel-trace --stats -d gpurun_out/abv -o r --output-format csv -- python bench.py $lean > gpurun_out/abv.log 2>&1
  python - <<PY
import pandas as pd
st = pd.read_csv('gpurun_out/abv/r_kernel_stats.csv')
st = st[st.Name.str.contains('render_kernel|physics_kernel')]
print('$lib:$impl', ' | '.join(f"{n.split('(')[0].split('::')[-1][:22]} avg {a/1e3:.1f} min {m/1e3:.1f} us" for n, a, m in zip(st.Name, st.AverageNs, st.MinNs)))
PY
done
