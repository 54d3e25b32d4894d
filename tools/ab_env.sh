#!/bin/bash
# usage: tools/ab_env.sh "VAR=value ..." "VAR=value ..."  -> bench.py under each environment setting ($BENCH_ARGS appended)
for e in "$@"; do
  env $e python bench.py --steps 60 --warmup 10 --no-cpu-baseline $BENCH_ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', 'step_ms', round(d['ms_per_step'],4), 'render_ms', round(d['roofline']['avg_launch_ms'],4), 'Msteps/s', round(d['value']/1e6,2))"
done
