#!/bin/bash
# Round 5, third GPU call: the suite with the full-diversity worlds (C2 4096 plans, C5 4096 large plans through 64-bit pool
# bases), the driver-shaped bench line with C5 on 4096 plans in `shapes`, and the wide colour instantiation with and without
# scratch (MS_WAVES_WIDE=5: 87 VGPRs, five waves a SIMD, no spill) at 512 rays and on C5's share.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c3; O=gpurun_out/c3
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider --durations=8 > $O/test.log 2>&1; echo "pytest rc=$?"; tail -16 $O/test.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; echo "bench rc=$?"; tail -4 $O/bench_k20.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/c3/bench_k20.json') if l.startswith('{')][-1])
    print('value', round(d['value']/1e6, 2), 'M  ms/step', round(d['ms_per_step'], 5), 'render', d['roofline']['avg_launch_ms'])
    for k, v in d['shapes'].items():
        print(f"{k:28s} ms/step {v['ms_per_step']:.4f} render {v['render_launch_ms']:.4f} plans {v['distinct_floorplans']} grid {v.get('wall_grid', {}).get('bytes', 0)/2**30:.2f} GiB cell {v.get('wall_grid', {}).get('cell')}")
    print(d['env_step_headline_shape'])
except Exception as e:
    print('no bench line', e)
PY
for lib in product w5; do
  if [ $lib = w5 ]; then export MEGASTEP_HIP_LIB=$PWD/megastep_amd/csrc/variants/w5.so; else unset MEGASTEP_HIP_LIB; fi
  echo "== $lib"; timeout 600 python tools/ab_groups.py --groups=1,4 r512 c5 2> $O/ab_$lib.err | tee $O/ab_$lib.txt
done
