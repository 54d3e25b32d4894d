#!/bin/bash
# Round 5, thirteenth GPU call: which kernels the reference-shaped demo envs' steps launch besides the two of the hot path.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c13; O=gpurun_out/c13
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in deathmatch explorer; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/t_$w -o r --output-format csv -- python tools/env_trace.py $w > $O/$w.log 2>&1
  python - $w <<'PY'
import pandas as pd, glob, sys
w = sys.argv[1]
f = glob.glob(f'gpurun_out/c13/t_{w}/**/r_kernel_stats.csv', recursive=True)[0]
st = pd.read_csv(f); st['us'] = st.AverageNs/1e3; st['per_step'] = st.Calls/110; st['us_per_step'] = st.TotalDurationNs/1e3/110
st['n'] = st.Name.str.replace('void ', '').str.replace('(anonymous namespace)::', '').str.slice(0, 95)
st = st[st.Calls >= 100].sort_values('TotalDurationNs', ascending=False)
print(w, 'kernels with >= 100 calls; total us per step', round(st.us_per_step.sum(), 1))
print(st[['n', 'Calls', 'us', 'us_per_step']].head(32).to_string())
st.to_csv(f'gpurun_out/c13/{w}_kernel_stats.csv', index=False)
PY
  rm -rf $O/t_$w
done
