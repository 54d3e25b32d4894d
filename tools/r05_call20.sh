#!/bin/bash
# Round 5, twentieth GPU call: the profiles the last kernel changes touch (physics' divisor: headline; the colourless mask: c2d) and the
# bench lines of record on the final build.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c20; O=gpurun_out/c20
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build_smoke.log 2>&1; echo "build+smoke rc=$?"; tail -1 $O/build_smoke.log
prof() { tag=$1; shift; timeout 1200 bash tools/profile.sh $tag "$@" > $O/profile_$tag.log 2>&1; echo "profile $tag rc=$?"; }
prof headline
prof c2d --agents 1 --depth-only
python - <<'PY'
import json
for t in ('headline', 'c2d'):
    e = json.load(open(f'gpurun_out/prof_{t}/traffic.json'))
    print(t, {k: round(v, 2) for k, v in e['kernel_us'].items()}, 'render MB', round(e['render_bytes_per_launch']/1e6, 1), 'busy', round(e['valu_busy_frac']['render_kernel'], 3), {k: round(v['VALU_per_wave'], 1) for k, v in e['per_wave'].items()})
PY
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; echo "bench k20 rc=$?"
timeout 600 python bench.py --no-shapes > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
