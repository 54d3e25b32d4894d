#!/bin/bash
# Round 6: the Deathmatch logic kernel - its tests, the env.step rates with it, a kernel trace of a fused and an unfused step.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6b; O=gpurun_out/r6b
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 600 python -m pytest tests/test_gpu_envs.py tests/test_abi.py -q --tb=short -p no:cacheprovider -x > $O/test.log 2>&1; echo "pytest rc=$?"; tail -15 $O/test.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-shapes > $O/bench_env.json 2> $O/bench_env.err; echo "bench rc=$?"; tail -3 $O/bench_env.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r6b/bench_env.json'))
print(json.dumps(d['env_step']['deathmatch']))
PY
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o dm --output-format csv -- python tools/env_trace.py deathmatch > $O/trace.log 2>&1
python - <<'PY'
import pandas as pd
st = pd.read_csv('gpurun_out/r6b/trace/dm_kernel_stats.csv')
st = st.sort_values('TotalDurationNs', ascending=False).head(25)
print(st[['Name', 'Calls', 'AverageNs', 'TotalDurationNs']].to_string(max_colwidth=70))
st.to_csv('gpurun_out/r6b/env_deathmatch_kernel_stats.csv', index=False)
PY
rm -rf $O/trace
