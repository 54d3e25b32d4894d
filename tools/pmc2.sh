#!/bin/bash
# usage: tools/pmc2.sh  -> what the SQ is busy with during render/physics: two PMC passes over python bench.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --kernel-trace -d gpurun_out/pmc2a -o p --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/pmc2a.log 2>&1
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS_ATOMIC SQ_LEVEL_WAVES SQ_IFETCH SQ_WAVE_CYCLES --kernel-trace -d gpurun_out/pmc2b -o p --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/pmc2b.log 2>&1
python - <<PY
import pandas as pd
for t in 'ab':
    d = pd.read_csv(f'gpurun_out/pmc2{t}/p_counter_collection.csv')
    d['k'] = d.Kernel_Name.str.extract(r'(render_kernel|render_prep_kernel|physics_kernel)')
    g = d[d.k.notna()].groupby(['k','Counter_Name']).Counter_Value.mean().unstack()
    print(g.round(0).T.to_string())
PY
