#!/bin/bash
# The round's rocprofv3 evidence for one bench shape, into gpurun_out/prof_<tag>/ (copy the summaries to profiles/):
# kernel-trace stats of the bench command (eager + graph legs), HBM traffic counters in separate PMC passes (FETCH_SIZE and
# WRITE_SIZE cannot share a pass on gfx950), two SQ counter sets - each PMC pass on the eager leg only.
# usage: tools/profile.sh <tag> [bench.py shape arguments, e.g. --res 128]
tag=$1; shift
shape="$@"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; mkdir -p $out
# (floorplans first, in a plain process that may fork its workers: the profiled runs load them - 4096 large plans are 90 s on one core)
python bench.py --plan-cache /tmp/plans_$tag.pkl --plans-only $shape > /dev/null 2> $out/plans.log
lean="--no-cpu-baseline --no-env-fps --no-shapes --plan-workers 0 --plan-cache /tmp/plans_$tag.pkl"
rocprofv3 --kernel-trace --stats -d $out/stats -o bench --output-format csv -- python bench.py --steps 100 --warmup 10 $lean $shape > $out/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o bench --output-format csv -- python bench.py --steps 10 --warmup 3 --no-graph $lean $shape > $out/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o bench --output-format csv -- python bench.py --steps 10 --warmup 3 --no-graph $lean $shape > $out/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -d $out/sq -o bench --output-format csv -- python bench.py --steps 10 --warmup 3 --no-graph $lean $shape > $out/bench_sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/active -o bench --output-format csv -- python bench.py --steps 10 --warmup 3 --no-graph $lean $shape > $out/bench_active.log 2>&1
python - "$out" "$tag" $shape <<'PY'
import json, os, re, sys
import pandas as pd
out, tag, shape = sys.argv[1], sys.argv[2], sys.argv[3:]
import argparse
ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=4096); ap.add_argument('--agents', type=int, default=4); ap.add_argument('--res', type=int, default=64)
ap.add_argument('--large', action='store_true'); ap.add_argument('--depth-only', action='store_true')
ap.add_argument('--unique', type=int, default=0); ap.add_argument('--legacy-plans', action='store_true'); ap.add_argument('--one-launch', action='store_true')
w, _ = ap.parse_known_args(shape)
# (the world's distinct floorplans, as bench.py counts them: a profile belongs to a shape AND a plan count)
plans = 460 if w.legacy_plans else max(1, min(w.unique, w.envs)) if w.unique else (w.envs if w.agents == 1 else max(w.envs//4, 1))
KERNELS = r'(render_kernel|render_prep_kernel|physics_kernel|dynlight_kernel)'
st = pd.read_csv(f'{out}/stats/bench_kernel_stats.csv')
st = st[st.Name.str.contains('render_kernel|render_prep|physics_kernel|dynlight|bake_kernel|bake_sum|visibility|lightgrid|lightlist|wallgrid')]
print(st[['Name', 'Calls', 'AverageNs', 'MinNs', 'MaxNs']].to_string())
st.to_csv(f'{out}/kernel_stats.csv', index=False)
res = {}
for c, f in [('FETCH_SIZE', 'fetch'), ('WRITE_SIZE', 'write')]:
    d = pd.read_csv(f'{out}/{f}/bench_counter_collection.csv')
    d['k'] = d.Kernel_Name.str.extract(KERNELS)
    g = d[d.k.notna() & (d.Counter_Name == c)].groupby('k').Counter_Value.mean()
    res[c] = g.to_dict(); print(c, '(KB per launch, raw counter)', g.round(0).to_dict())
# HBM bytes per launch: FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read;
# calibrated here on physics_kernel's 16 B/lane wall stream in round 2), WRITE_SIZE (KB) as reported.
rb = sum(2*1024*res['FETCH_SIZE'].get(k, 0) + 1024*res['WRITE_SIZE'].get(k, 0) for k in ('render_kernel', 'render_prep_kernel', 'dynlight_kernel'))
pb = 2*1024*res['FETCH_SIZE'].get('physics_kernel', 0) + 1024*res['WRITE_SIZE'].get('physics_kernel', 0)
traffic = {'workload': {'envs': w.envs, 'agents': w.agents, 'res': w.res, 'large': w.large, 'depth_only': w.depth_only, 'plans': plans, **({'one_launch': True} if w.one_launch else {})}, 'shape': tag,
           'render_bytes_per_launch': rb, 'physics_bytes_per_launch': pb, 'raw_counters_KB': res,
           'kernel_us': {re.search(r'((?:render|physics)_kernel<[^>]*>)', r.Name).group(1): r.AverageNs/1e3
                         for r in st.itertuples() if re.search(r'(?:render|physics)_kernel<', r.Name)},
           'method': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on python bench.py --no-graph ' + ' '.join(shape)
                     + '; bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024'}
sq = pd.read_csv(f'{out}/sq/bench_counter_collection.csv')
sq['k'] = sq.Kernel_Name.str.extract(KERNELS)
g = sq[sq.k.notna()].groupby(['k', 'Counter_Name']).Counter_Value.mean().unstack()
for c in ('VALU', 'SALU', 'LDS'):
    g[f'{c}_per_wave'] = g[f'SQ_INSTS_{c}']/g.SQ_WAVES
# (SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_* count quad-cycles: x4 for shader cycles)
g['cycles_per_wave'] = 4*g.SQ_WAVE_CYCLES/g.SQ_WAVES
g['parked_frac'] = g.SQ_WAIT_ANY/g.SQ_WAVE_CYCLES
g['issue_stall_frac'] = g.SQ_WAIT_INST_ANY/g.SQ_WAVE_CYCLES
g.to_csv(f'{out}/sq_counters_mean_per_launch.csv')
print(g.round(2).to_string())
f = f'{out}/active/bench_counter_collection.csv'
if os.path.exists(f):
    # How busy the vector ALUs are: SQ_ACTIVE_INST_VALU counts quad-cycles (one per wave64 instruction's four cycles), summed
    # over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE is the launch's length in shader cycles, summed over the 8 XCDs.
    d = pd.read_csv(f)
    d['k'] = d.Kernel_Name.str.extract(r'(render_kernel|physics_kernel)')
    a = d[d.k.notna()].groupby(['k', 'Counter_Name']).Counter_Value.mean().unstack()
    a['cycles_per_valu_inst'] = 4*a.SQ_ACTIVE_INST_VALU/a.SQ_INSTS_VALU
    a['valu_busy_frac'] = 4*a.SQ_ACTIVE_INST_VALU/1024/(a.GRBM_GUI_ACTIVE/8)
    a.to_csv(f'{out}/valu_busy.csv'); print(a.round(3).to_string())
    traffic['valu_busy_frac'] = a.valu_busy_frac.to_dict()
traffic['per_wave'] = {k: {c: float(g.loc[k, c]) for c in ('VALU_per_wave', 'SALU_per_wave', 'LDS_per_wave', 'cycles_per_wave', 'parked_frac', 'issue_stall_frac')}
                       for k in g.index}
json.dump(traffic, open(f'{out}/traffic.json', 'w'), indent=1)
PY
tail -1 $out/bench_stats.log | cut -c1-400
# (gpurun brings back at most 64 MiB: the raw traces stay on the box, the summaries above travel)
cp $out/stats/bench_kernel_stats.csv $out/kernel_stats_all.csv 2>/dev/null
rm -rf $out/stats $out/fetch $out/write $out/sq $out/active
