#!/bin/bash
# usage: tools/ab.sh lib1.so lib2.so ... ; prints render ms for each
for lib in "$@"; do
  MEGASTEP_HIP_LIB=$PWD/$lib python bench.py --steps 100 --warmup 10 --no-cpu-baseline $BENCH_ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', 'step_ms', round(d['ms_per_step'],4), 'render_ms', round(d['roofline']['avg_launch_ms'],4), 'Msteps/s', round(d['value']/1e6,2))"
done
