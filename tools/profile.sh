#!/bin/bash
# Collects the round's rocprofv3 evidence into gpurun_out/prof_$1/: kernel-trace stats of the default bench command
# (eager + graph legs), then HBM traffic counters in separate PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass
# on gfx950) and an SQ counter set, each on the eager leg only.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; mkdir -p $out
lean="--no-cpu-baseline --no-env-fps"
rocprofv3 --kernel-trace --stats -d $out/stats -o bench --output-format csv -- python bench.py --steps 100 --warmup 10 $lean > $out/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o bench --output-format csv -- python bench.py --steps 10 --warmup 3 --no-graph $lean > $out/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o bench --output-format csv -- python bench.py --steps 10 --warmup 3 --no-graph $lean > $out/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -d $out/sq -o bench --output-format csv -- python bench.py --steps 10 --warmup 3 --no-graph $lean > $out/bench_sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/active -o bench --output-format csv -- python bench.py --steps 10 --warmup 3 --no-graph $lean > $out/bench_active.log 2>&1
python - <<PY
import pandas as pd, json
out='$out'
st = pd.read_csv(f'{out}/stats/bench_kernel_stats.csv')
st = st[st.Name.str.contains('render_kernel|render_prep|physics_kernel|dynlight|bake_kernel|bake_sum|visibility|lightgrid|lightlist')]
print(st[['Name','Calls','AverageNs','MinNs','MaxNs']].to_string())
st.to_csv(f'{out}/kernel_stats.csv', index=False)
res = {}
for c, f in [('FETCH_SIZE','fetch'),('WRITE_SIZE','write')]:
    d = pd.read_csv(f'{out}/{f}/bench_counter_collection.csv')
    d['k'] = d.Kernel_Name.str.extract(r'(render_kernel|render_prep_kernel|physics_kernel|dynlight_kernel)')
    g = d[d.k.notna() & (d.Counter_Name==c)].groupby('k').Counter_Value.mean()
    res[c] = g.to_dict(); print(c, '(KB per launch, raw counter)', g.round(0).to_dict())
json.dump(res, open(f'{out}/traffic_raw.json','w'))
# HBM bytes per ms_render launch: FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide
# coalesced read; calibrated here on physics_kernel's 16 B/lane wall stream), WRITE_SIZE (KB) as reported.
rb = sum(2*1024*res['FETCH_SIZE'].get(k, 0) + 1024*res['WRITE_SIZE'].get(k, 0) for k in ('render_kernel', 'render_prep_kernel', 'dynlight_kernel'))
pb = 2*1024*res['FETCH_SIZE'].get('physics_kernel', 0) + 1024*res['WRITE_SIZE'].get('physics_kernel', 0)
json.dump({'workload': {'envs': 4096, 'agents': 4, 'res': 64, 'large': False},
           'render_bytes_per_launch': rb, 'physics_bytes_per_launch': pb, 'raw_counters_KB': res,
           'method': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on python bench.py --no-graph; bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024'},
          open(f'{out}/traffic.json','w'), indent=1)
sq = pd.read_csv(f'{out}/sq/bench_counter_collection.csv')
sq['k'] = sq.Kernel_Name.str.extract(r'(render_kernel|render_prep_kernel|physics_kernel|dynlight_kernel)')
g = sq[sq.k.notna()].groupby(['k','Counter_Name']).Counter_Value.mean().unstack()
g['VALU_per_wave'] = g.SQ_INSTS_VALU/g.SQ_WAVES; g['SALU_per_wave'] = g.SQ_INSTS_SALU/g.SQ_WAVES; g['LDS_per_wave'] = g.SQ_INSTS_LDS/g.SQ_WAVES
g.to_csv(f'{out}/sq_counters_mean_per_launch.csv')
print(g.round(0).to_string())
PY
python - <<PY
# How busy the vector ALUs are: SQ_ACTIVE_INST_VALU counts cycles (4 per wave64 instruction), summed over the chip's
# 1024 SIMDs; GRBM_GUI_ACTIVE is the launch's length in shader cycles (summed over the 8 XCDs).
import pandas as pd, json, os
out='$out'
f = f'{out}/active/bench_counter_collection.csv'
if os.path.exists(f):
    d = pd.read_csv(f)
    d['k'] = d.Kernel_Name.str.extract(r'(render_kernel|physics_kernel)')
    g = d[d.k.notna()].groupby(['k','Counter_Name']).Counter_Value.mean().unstack()
    g['cycles_per_valu_inst'] = g.SQ_ACTIVE_INST_VALU/g.SQ_INSTS_VALU
    g['valu_busy_frac'] = g.SQ_ACTIVE_INST_VALU/1024/(g.GRBM_GUI_ACTIVE/8)
    g.to_csv(f'{out}/valu_busy.csv'); print(g.round(3).to_string())
PY
tail -1 $out/bench_stats.log | cut -c1-600
