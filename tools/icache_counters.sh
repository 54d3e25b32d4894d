#!/bin/bash
# The instruction cache's view of one bench shape: requests, hits, misses of the render and physics kernels per launch (one
# rocprofv3 --pmc pass, eager leg, 10 steps).   usage: tools/icache_counters.sh <tag> [bench.py shape arguments]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ic_$tag; mkdir -p $out
timeout 150 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $out/p -o bench --output-format csv -- python bench.py --no-cpu-baseline --no-env-fps --no-shapes --plan-workers 0 --steps 10 --warmup 3 --no-graph "$@" > $out/log.txt 2>&1
python - "$out" "$tag" <<'PY'
import sys, pandas as pd
out, tag = sys.argv[1:3]
d = pd.read_csv(f'{out}/p/bench_counter_collection.csv')
d['k'] = d.Kernel_Name.str.extract(r'(render_kernel|physics_kernel)')
g = d[d.k.notna()].groupby(['k', 'Counter_Name']).Counter_Value.mean().unstack()
g['miss_rate'] = g.SQC_ICACHE_MISSES/g.SQC_ICACHE_REQ
g['misses_per_wave'] = g.SQC_ICACHE_MISSES/g.SQ_WAVES
g['req_per_wave'] = g.SQC_ICACHE_REQ/g.SQ_WAVES
print(tag); print(g.round(3).T.to_string())
g.to_csv(f'{out}/icache.csv')
PY
rm -rf $out/p
