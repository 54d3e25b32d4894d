#!/bin/bash
# Round 6: ms_step_render (single-agent worlds of <= 64 rays as one launch a step) - tests, then C2 both ways under the bench protocol.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6c; O=gpurun_out/r6c
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_gpu_step_render.py tests/test_gpu_parity.py -q --tb=short -p no:cacheprovider -x > $O/test.log 2>&1; echo "pytest rc=$?"; tail -25 $O/test.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-env-fps > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r6c/bench.json'))
print('headline', d['value'], d['ms_per_step'])
for k, v in d['shapes'].items():
    if k.startswith('c2'):
        print(k, round(v['ms_per_step']*1e3, 2), 'us/step', round(v['env_steps_per_s']/1e6, 1), 'M/s; render event', round(v['render_launch_ms']*1e3, 2), 'launches', v.get('launches_per_step', 2))
PY
