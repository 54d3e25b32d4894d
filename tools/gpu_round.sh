#!/bin/bash
# One gpurun call's worth of checks: the GPU test suite, the default bench line, the rocprofv3 evidence.
# usage: tools/gpu_round.sh <tag> [pytest args...]
tag=$1; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_$tag.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider "$@" > gpurun_out/test_$tag.log 2>&1
echo "pytest rc=$?" >> gpurun_out/test_$tag.log
tail -25 gpurun_out/test_$tag.log
timeout 600 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
echo "bench rc=$?"; cut -c1-1500 gpurun_out/bench_$tag.json; tail -5 gpurun_out/bench_$tag.err
timeout 900 bash tools/profile.sh $tag
