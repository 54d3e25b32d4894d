#!/bin/bash
# Round 6, first GPU call: the suite with the oblique-floorplan tests, the default bench line (new shapes: headline on 4096 plans,
# headline on oblique plans; env.step legs on one plan per core env), 8 ranks sharing the one GPU (N > 1 plumbing on real HIP).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6a; O=gpurun_out/r6a
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $O/test.log 2>&1; echo "pytest rc=$?"; tail -8 $O/test.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench rc=$?"; cut -c1-600 $O/bench_steps20.json; tail -3 $O/bench_steps20.err
timeout 600 python bench.py --gpus 8 --share-gpu --steps 20 --warmup 5 --no-cpu-baseline --no-env-fps --no-shapes > $O/bench_8ranks_share.json 2> $O/bench_8ranks_share.err; echo "share rc=$?"; cut -c1-400 $O/bench_8ranks_share.json; tail -3 $O/bench_8ranks_share.err
timeout 300 python tools/fuzz_parity.py 200 60 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
