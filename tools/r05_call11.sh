#!/bin/bash
# Round 5, eleventh GPU call: the round's rocprofv3 evidence per shape on the final kernels, and the bench lines of record.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c11; O=gpurun_out/c11
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
prof() { tag=$1; shift; timeout 1200 bash tools/profile.sh $tag "$@" > $O/profile_$tag.log 2>&1; echo "profile $tag rc=$?"; }
prof headline
prof c2 --agents 1
prof c2d --agents 1 --depth-only
prof c3 --res 128 --fov 70
prof r512 --res 512 --fov 70
prof c5 --envs 32768 --agents 1 --res 256 --large --unique 4096 --fast-build
python - <<'PY'
import json
for t in ('headline', 'c2', 'c2d', 'c3', 'r512', 'c5'):
    e = json.load(open(f'gpurun_out/prof_{t}/traffic.json'))
    print(t, {k: round(v, 1) for k, v in e['kernel_us'].items()}, 'render MB', round(e['render_bytes_per_launch']/1e6, 1), 'busy', round(e['valu_busy_frac']['render_kernel'], 3), 'VALU/wave', round(e['per_wave']['render_kernel']['VALU_per_wave'], 1), 'SALU', round(e['per_wave']['render_kernel']['SALU_per_wave'], 1))
PY
timeout 600 python bench.py --no-shapes > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; echo "bench k20 rc=$?"
