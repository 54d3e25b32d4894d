#!/bin/bash
# Round 5, first GPU call: the suite on the round's first changes, the driver-shaped bench line, C5's share on the reference's plan
# diversity (4096 large plans instead of 64) under two wall-grid budgets, and a kernel trace of env.step at the headline shape.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c1; O=gpurun_out/c1
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/test.log 2>&1; echo "pytest rc=$?"; tail -5 $O/test.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; echo "bench rc=$?"; cut -c1-600 $O/bench_k20.json
C5="--envs 32768 --agents 1 --res 256 --large --fast-build --steps 20 --warmup 5 --no-cpu-baseline --no-env-fps --no-shapes"
export MEGASTEP_VERBOSE=1
timeout 600 python bench.py $C5 --unique 64 > $O/c5_u64.json 2> $O/c5_u64.err; echo "c5 u64 rc=$?"
timeout 900 python bench.py $C5 --unique 4096 > $O/c5_u4096.json 2> $O/c5_u4096.err; echo "c5 u4096 rc=$?"
MEGASTEP_WALL_GRID_BYTES=100e9 timeout 900 python bench.py $C5 --unique 4096 > $O/c5_u4096_big.json 2> $O/c5_u4096_big.err; echo "c5 u4096 big rc=$?"
grep -h "megastep_amd:" $O/c5_*.err
for f in c5_u64 c5_u4096 c5_u4096_big; do python - $O/$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[1], 'ms/step', round(d['ms_per_step'], 4), 'render ms', round(d['roofline']['avg_launch_ms'], 4), 'value', round(d['value']/1e6, 1), 'M', d['config'].get('wall_grid'))
except Exception as e:
    print(sys.argv[1], 'no line', e)
PY
done
unset MEGASTEP_VERBOSE
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $O/envk -o r --output-format csv -- python tools/env_step_kernels.py > $O/envk.log 2>&1
python - <<'PY'
import pandas as pd, glob
f = glob.glob('gpurun_out/c1/envk/**/r_kernel_stats.csv', recursive=True)
if f:
    st = pd.read_csv(f[0]); st['us'] = st.AverageNs/1e3
    print(st[st.Calls >= 100][['Name', 'Calls', 'us', 'TotalDurationNs']].sort_values('TotalDurationNs', ascending=False).head(30).to_string())
    st.to_csv('gpurun_out/c1/envk_kernel_stats.csv', index=False)
PY
tail -2 $O/envk.log | cut -c1-600
rm -rf $O/envk
