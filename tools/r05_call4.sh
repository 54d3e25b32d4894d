#!/bin/bash
# Round 5, fourth GPU call: suite + fuzz on the build with 64-bit pool bases, the round's rocprofv3 evidence per shape, bench lines.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c4; O=gpurun_out/c4
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/test.log 2>&1; echo "pytest rc=$?"; tail -3 $O/test.log
timeout 900 python tools/fuzz_parity.py 5000 250 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; echo "bench k20 rc=$?"
timeout 600 python bench.py --no-shapes > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
python - <<'PY'
import json
for f in ('bench_k20', 'bench_default'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/c4/{f}.json') if l.startswith('{')][-1])
        print(f, 'value', round(d['value']/1e6, 2), 'M  ms/step', round(d['ms_per_step'], 5), 'render', round(d['roofline']['avg_launch_ms'], 5), d['env_step_headline_shape'])
        if 'env_step' in d: print({k: (round(v['fps']/1e6, 1), round(v['fps_hip_graph']/1e6, 1)) for k, v in d['env_step'].items()})
    except Exception as e:
        print(f, 'no line', e)
PY
prof() { tag=$1; shift; timeout 1200 bash tools/profile.sh $tag "$@" > $O/profile_$tag.log 2>&1; echo "profile $tag rc=$?"; grep -E "render_kernel|physics_kernel" $O/profile_$tag.log | head -4 | cut -c1-200; }
prof headline
prof c2 --agents 1
prof c2d --agents 1 --depth-only
prof c3 --res 128 --fov 70
prof r512 --res 512 --fov 70
prof c5 --envs 32768 --agents 1 --res 256 --large --unique 4096 --fast-build
ls gpurun_out/prof_*/traffic.json
