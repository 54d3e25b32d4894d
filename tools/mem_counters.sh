#!/bin/bash
# The memory pipeline's view of one bench shape: texture-address unit, vector L1, address translation and L2 counters of
# the render and physics kernels, one rocprofv3 --pmc pass per group (eager leg, 10 steps), means per launch.
# usage: tools/mem_counters.sh <tag> [bench.py shape arguments]
tag=$1; shift
shape="$@"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/mem_$tag; mkdir -p $out
lean="--no-cpu-baseline --no-env-fps --no-shapes --plan-workers 0 --steps 10 --warmup 3 --no-graph"
i=0
for group in "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" \
             "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_GATE_EN1_sum" \
             "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $group --kernel-trace -d $out/p$i -o bench --output-format csv -- python bench.py $lean $shape > $out/p$i.log 2>&1
done
python - "$out" <<'PY'
import glob, sys
import pandas as pd
out = sys.argv[1]
rows = []
for f in sorted(glob.glob(f'{out}/p*/bench_counter_collection.csv')):
    d = pd.read_csv(f)
    d['k'] = d.Kernel_Name.str.extract(r'(render_kernel|physics_kernel)')
    rows.append(d[d.k.notna()].groupby(['k', 'Counter_Name']).Counter_Value.mean())
g = pd.concat(rows).unstack(0)
pd.set_option('display.float_format', lambda v: f'{v:,.1f}')
print(g.to_string())
g.to_csv(f'{out}/mem_counters_mean_per_launch.csv')
PY
rm -rf $out/p[0-9]
