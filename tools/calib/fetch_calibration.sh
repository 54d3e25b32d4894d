#!/bin/bash
# FETCH_SIZE against known byte counts, per access pattern (tools/calib/gather_bench.hip).  Run on an MI355X (gpurun).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/calib; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $out/gather_bench tools/calib/gather_bench.hip || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/f -o g --output-format csv -- $out/gather_bench > $out/f.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace -d $out/r -o g --output-format csv -- $out/gather_bench > $out/r.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/s -o g --output-format csv -- $out/gather_bench > $out/s.log 2>&1
python - "$out" <<'PY'
import json, sys
import pandas as pd
out = sys.argv[1]
useful = {'stream16': 16, 'gather4': 4, 'gather12': 12, 'gather16': 16}
lanes = 1 << 24
rows = {}
f = pd.read_csv(f'{out}/f/g_counter_collection.csv'); f['k'] = f.Kernel_Name.str.extract(r'(stream16|gather4|gather12|gather16)')
fetch = f[f.k.notna()].groupby('k').Counter_Value.mean()
r = pd.read_csv(f'{out}/r/g_counter_collection.csv'); r['k'] = r.Kernel_Name.str.extract(r'(stream16|gather4|gather12|gather16)')
req = r[r.k.notna()].groupby(['k', 'Counter_Name']).Counter_Value.mean().unstack()
st = pd.read_csv(f'{out}/s/g_kernel_stats.csv'); st['k'] = st.Name.str.extract(r'(stream16|gather4|gather12|gather16)')
dur = st[st.k.notna()].set_index('k').AverageNs
for k, b in useful.items():
    ub = lanes*b
    q = req.loc[k]
    n32, n64, n128 = q.get('TCC_EA0_RDREQ_32B_sum', 0), q.get('TCC_EA0_RDREQ_64B_sum', 0), q.get('TCC_EA0_RDREQ_128B_sum', 0)
    other = q['TCC_EA0_RDREQ_sum'] - n32 - n64 - n128
    rows[k] = dict(useful_bytes=ub, FETCH_SIZE_KB=float(fetch[k]), fetch_bytes_over_useful=float(fetch[k])*1024/ub,
                   rdreq=float(q['TCC_EA0_RDREQ_sum']), rdreq_32B=float(n32), rdreq_64B=float(n64), rdreq_128B=float(n128), rdreq_other=float(other),
                   rdreq_per_lane=float(q['TCC_EA0_RDREQ_sum'])/lanes, avg_us=float(dur[k])/1e3, useful_GBps=ub/float(dur[k]))
print(pd.DataFrame(rows).T.round(3).to_string())
json.dump(rows, open(f'{out}/fetch_calibration.json', 'w'), indent=1)
PY
rm -rf $out/f $out/r $out/s $out/gather_bench
