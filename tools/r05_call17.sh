#!/bin/bash
# Round 5, seventeenth GPU call: optional outputs asked of a resident mask instead of kernel-argument loads: suite, fuzz, A/B, env-step rates.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c17; O=gpurun_out/c17
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/test.log 2>&1; echo "pytest rc=$?"; tail -3 $O/test.log
timeout 900 python tools/fuzz_parity.py 14000 300 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
timeout 600 python tools/ab_envstep.py --res 512 --fov 70 --sub 4 --centre 2> $O/err.txt | tee $O/ab_obs512.txt
timeout 600 python tools/ab_envstep.py --agents 1 --res 256 --sub 4 2>> $O/err.txt | tee $O/ab_obs256.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-shapes --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/c17/bench.json') if l.startswith('{')][-1])
print('value', round(d['value']/1e6, 2), 'M', round(d['ms_per_step'], 5)); e = d['env_step_headline_shape']; print(round(e['ms_per_step']*1e3, 1), round(e['ms_per_step_hip_graph']*1e3, 1), 'us')
print({k: (round(v['fps']/1e6, 1), round(v['fps_hip_graph']/1e6, 1)) for k, v in d['env_step'].items()})
PY
