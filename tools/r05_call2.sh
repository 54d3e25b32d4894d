#!/bin/bash
# Round 5, second GPU call: (1) env.step's two launches against the bench's, variant by variant; (2) C5's share against the number
# of distinct floorplans at a fixed 0.25 m wall grid (what is cache, what is cell size), and at 0.5 m on 64 plans.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c2; O=gpurun_out/c2
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 600 python tools/ab_envstep.py > $O/ab_envstep.txt 2> $O/ab_envstep.err; echo "ab_envstep rc=$?"; cat $O/ab_envstep.txt; tail -3 $O/ab_envstep.err
C5="--envs 32768 --agents 1 --res 256 --large --fast-build --steps 20 --warmup 5 --no-cpu-baseline --no-env-fps --no-shapes"
export MEGASTEP_WALL_GRID_BYTES=60e9
run() {
    tag=$1; shift
    timeout 900 python bench.py $C5 "$@" > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; grep -h "megastep_amd:" $O/$tag.err
    python - $O/$tag.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    wg = d['config'].get('wall_grid', {})
    print(sys.argv[1], 'ms/step', round(d['ms_per_step'], 4), 'render ms', round(d['roofline']['avg_launch_ms'], 4), 'eager ms/step', round(d['eager']['ms_per_step'], 4), 'value', round(d['value']/1e6, 1), 'M; grid', round(wg.get('bytes', 0)/2**30, 2), 'GiB cell', wg.get('cell'))
except Exception as e:
    print(sys.argv[1], 'no line', e)
PY
}
MEGASTEP_WALL_GRID_CELL=0.5 run c5_u64_cell05 --unique 64
run c5_u64 --unique 64
run c5_u512 --unique 512
run c5_u1024 --unique 1024
run c5_u2048 --unique 2048
