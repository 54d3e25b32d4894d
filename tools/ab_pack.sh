#!/bin/bash
# physics_kernel's PACK (envs side by side per wave) A/B: kernel times (rocprofv3 --kernel-trace --stats) and ms/step per
# setting of MEGASTEP_PHYSICS_PACK and bench shape.   usage: tools/ab_pack.sh "1 0 4 8" "--envs 32768 --agents 1 ..." [...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
packs=$1; shift
lean="--no-cpu-baseline --no-env-fps --no-shapes --plan-workers 0 --steps 40 --warmup 5"
for shape in "$@"; do for k in $packs; do
  MEGASTEP_PHYSICS_PACK=$k timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/abp -o r --output-format csv -- python bench.py $lean $shape > gpurun_out/abp.log 2> gpurun_out/abp.err
  python - "$k" "$shape" <<'PY'
import pandas as pd, re, sys
st = pd.read_csv('gpurun_out/abp/r_kernel_stats.csv')
st = st[st.Name.str.contains('render_kernel|physics_kernel')]
ms = re.search(r'"ms_per_step": ([0-9.]+)', open('gpurun_out/abp.log').read())
print('pack', sys.argv[1], repr(sys.argv[2]), ' | '.join('%s avg %.1f min %.1f us' % (re.search(r'(\w+_kernel)', n).group(1), a/1e3, m/1e3) for n, a, m in zip(st.Name, st.AverageNs, st.MinNs)),
      '| step', ms.group(1) if ms else None, 'ms')
PY
done; done
