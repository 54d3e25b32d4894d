#!/bin/bash
# Round 6: the Explorer books kernel - tests, env.step rates, a kernel trace of an Explorer step.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6e; O=gpurun_out/r6e
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 600 python -m pytest tests/test_gpu_envs.py tests/test_abi.py -q --tb=short -p no:cacheprovider -x > $O/test.log 2>&1; echo "pytest rc=$?"; tail -15 $O/test.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-shapes > $O/bench_env.json 2> $O/bench_env.err; echo "bench rc=$?"; tail -3 $O/bench_env.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r6e/bench_env.json'))
for k, v in d['env_step'].items(): print(k, round(v['fps']/1e6, 1), round(v['fps_hip_graph']/1e6, 1))
print(d['env_step_headline_shape']['env_steps_per_s'], d['env_step_headline_shape']['env_steps_per_s_hip_graph'])
PY
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o ex --output-format csv -- python tools/env_trace.py explorer > $O/trace.log 2>&1
python - <<'PY'
import pandas as pd
st = pd.read_csv('gpurun_out/r6e/trace/ex_kernel_stats.csv')
st = st[st.Calls >= 100].sort_values('TotalDurationNs', ascending=False).head(14)
print(st[['Name', 'Calls', 'AverageNs']].to_string(max_colwidth=80))
st.to_csv('gpurun_out/r6e/env_explorer_kernel_stats.csv', index=False)
PY
rm -rf $O/trace
python tools/env_host_profile.py 2>&1 | head -45 > $O/host_profile.txt; head -3 $O/host_profile.txt
