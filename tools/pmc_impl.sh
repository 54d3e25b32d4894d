#!/bin/bash
# SQ counters of the render kernel per MEGASTEP_RENDER_IMPL (two PMC passes each; kernel trace only). usage: tools/pmc_impl.sh "pairs v2"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
lean="--no-cpu-baseline --no-env-fps --no-graph --steps 8 --warmup 2"
for impl in $1; do
  MEGASTEP_RENDER_IMPL=$impl rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -d gpurun_out/pmcA_$impl -o p --output-format csv -- python bench.py $lean > gpurun_out/pmc_$impl.log 2>&1
  MEGASTEP_RENDER_IMPL=$impl rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d gpurun_out/pmcB_$impl -o p --output-format csv -- python bench.py $lean >> gpurun_out/pmc_$impl.log 2>&1
  python - <<PY
import pandas as pd
for t in 'AB':
    d = pd.read_csv('gpurun_out/pmc'+t+'_$impl/p_counter_collection.csv')
    d = d[d.Kernel_Name.str.contains('render_kernel')]
    g = d.groupby('Counter_Name').Counter_Value.mean()
    print('$impl', t, g.round(0).to_dict())
PY
done
