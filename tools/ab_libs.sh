#!/bin/bash
# Kernel times (rocprofv3 --kernel-trace --stats) and the bench's ms/step per library build and bench shape.
# usage: tools/ab_libs.sh "libmegastep_hip variants/w7 ..." ["" "--envs 16384" "--res 128" ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
libs=$1; shift
[ $# -eq 0 ] && set -- ""
lean="--no-cpu-baseline --no-env-fps --no-shapes --plan-workers 0 --steps 60 --warmup 5"
for shape in "$@"; do for lib in $libs; do
  MEGASTEP_HIP_LIB=$PWD/megastep_amd/csrc/$lib.so timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ab -o r --output-format csv -- python bench.py $lean $shape > gpurun_out/ab.log 2> gpurun_out/ab.err
  python - "$lib" "$shape" <<'PY'
import pandas as pd, re, sys
st = pd.read_csv('gpurun_out/ab/r_kernel_stats.csv')
st = st[st.Name.str.contains('render_kernel|physics_kernel')]
ms = re.search(r'"ms_per_step": ([0-9.]+)', open('gpurun_out/ab.log').read())
print(sys.argv[1], repr(sys.argv[2]), ' | '.join('%s avg %.1f min %.1f us' % (re.search(r'(\w+_kernel)', n).group(1), a/1e3, m/1e3) for n, a, m in zip(st.Name, st.AverageNs, st.MinNs)),
      '| step', ms.group(1) if ms else None, 'ms')
PY
done; done
