#!/bin/bash
# usage: tools/env_prof.sh  -> rocprofv3 kernel stats of whole env.step() loops (Explorer, Deathmatch), top kernels by time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/envprof -o e --output-format csv -- python tools/env_fps.py > gpurun_out/envprof.log 2>&1
tail -3 gpurun_out/envprof.log
python - <<PY
import pandas as pd
d = pd.read_csv('gpurun_out/envprof/e_kernel_stats.csv')
d['Name'] = d.Name.str.slice(0, 110)
print(d[['Name','Calls','TotalDurationNs','AverageNs','Percentage']].head(32).to_string())
PY
