#!/bin/bash
# Round 5, tenth GPU call: the suite and the fuzz on the build with one chunk of rows in flight (MS_AHEAD=1), bench line.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c10; O=gpurun_out/c10
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/test.log 2>&1; echo "pytest rc=$?"; tail -3 $O/test.log
timeout 900 python tools/fuzz_parity.py 7000 400 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
for g in 4 2; do MEGASTEP_RAY_GROUPS=$g timeout 600 python tools/fuzz_parity.py 7400 100 > $O/fuzz_g$g.log 2>&1; echo "fuzz groups $g rc=$?"; tail -1 $O/fuzz_g$g.log; done
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; echo "bench k20 rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/c10/bench_k20.json') if l.startswith('{')][-1])
print('value', round(d['value']/1e6, 2), 'M  ms/step', round(d['ms_per_step'], 5), 'render', round(d['roofline']['avg_launch_ms'], 5), 'frac', round(d['roofline']['frac'], 4))
for k, v in d['shapes'].items():
    print(f"{k:28s} ms/step {v['ms_per_step']:.4f} render {v['render_launch_ms']:.4f} frac {v['roofline_frac']:.3f} measured {v.get('frac_measured')} busy {v.get('valu_busy')}")
e = d['env_step_headline_shape']; print(round(e['ms_per_step'], 5), round(e['ms_per_step_hip_graph'], 5)); print({k: (round(v['fps']/1e6, 1), round(v['fps_hip_graph']/1e6, 1)) for k, v in d['env_step'].items()})
PY
