#!/bin/bash
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d gpurun_out/pmc2_$tag -o p --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/pmc2_$tag.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmc3_$tag -o p --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/pmc3_$tag.log 2>&1
python - <<PY
import pandas as pd
for f in ['pmc2_$tag','pmc3_$tag']:
    d = pd.read_csv(f'gpurun_out/{f}/p_counter_collection.csv')
    d = d[d.Kernel_Name.str.contains('render_kernel')]
    d['dur'] = d.End_Timestamp - d.Start_Timestamp
    g = d.groupby('Counter_Name').Counter_Value.mean()
    print('$tag', f, 'dur_us', round(d.dur.mean()/1e3,1)); print(g.apply(lambda v: f'{v:.4g}').to_string())
PY
