#!/bin/bash
# SQ counters per wave of the render kernel for one bench shape, quickly (one PMC pass): tools/sq_quick.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/sq_$tag; mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $out/p -o bench --output-format csv -- python bench.py --no-cpu-baseline --no-env-fps --no-shapes --plan-workers 0 --steps 10 --warmup 3 --no-graph "$@" > $out/log.txt 2>&1
python - "$out" "$tag" <<'PY'
import sys, pandas as pd
out, tag = sys.argv[1:3]
d = pd.read_csv(f'{out}/p/bench_counter_collection.csv')
d['k'] = d.Kernel_Name.str.extract(r'(render_kernel|physics_kernel)')
g = d[d.k.notna()].groupby(['k', 'Counter_Name']).Counter_Value.mean().unstack()
for c in ('VALU', 'SALU', 'LDS'):
    g[f'{c}_per_wave'] = g[f'SQ_INSTS_{c}']/g.SQ_WAVES
g['parked'] = g.SQ_WAIT_ANY/g.SQ_WAVE_CYCLES
g['issue_stall'] = g.SQ_WAIT_INST_ANY/g.SQ_WAVE_CYCLES
g['cycles_per_wave'] = 4*g.SQ_WAVE_CYCLES/g.SQ_WAVES
print(tag); print(g[['SQ_WAVES', 'SQ_INSTS_VALU', 'VALU_per_wave', 'SALU_per_wave', 'LDS_per_wave', 'cycles_per_wave', 'parked', 'issue_stall']].round(2).to_string())
g.to_csv(f'{out}/sq.csv')
PY
rm -rf $out/p
