#!/bin/bash
# usage: tools/pmc.sh lib.so tag  -> prints per-kernel mean counters
lib=$1; tag=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MEGASTEP_HIP_LIB=$PWD/$lib rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace -d gpurun_out/pmc_$tag -o p --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline $BENCH_ARGS > gpurun_out/pmc_$tag.log 2>&1
python - <<PY
import pandas as pd
d = pd.read_csv('gpurun_out/pmc_$tag/p_counter_collection.csv')
d['k'] = d.Kernel_Name.str.extract(r'(render_kernel|render_prep_kernel|physics_kernel|dynlight_kernel)')
g = d[d.k.notna()].groupby(['k','Counter_Name']).Counter_Value.mean().unstack()
g['VALU/wave'] = g.SQ_INSTS_VALU/g.SQ_WAVES; g['SALU/wave']=g.SQ_INSTS_SALU/g.SQ_WAVES; g['LDS/wave']=g.SQ_INSTS_LDS/g.SQ_WAVES; g['cyc/wave']=4*g.SQ_WAVE_CYCLES/g.SQ_WAVES
print('$tag'); print(g[['VALU/wave','SALU/wave','LDS/wave','cyc/wave']].round(0))
PY
