#!/bin/bash
# Round 5, last GPU call: what the driver will run, on HEAD - the GPU suite, smoke, the bench line.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c24; O=gpurun_out/c24
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build_smoke.log 2>&1; echo "build+smoke rc=$?"; tail -1 $O/build_smoke.log
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/test.log 2>&1; echo "pytest rc=$?"; tail -2 $O/test.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/c24/bench.json').read()); print(round(d['value']/1e6,2), 'M', d['ms_per_step'], len(d['shapes']), 'shapes', d['roofline']['frac'], d['cpu_baseline']['value'])"
