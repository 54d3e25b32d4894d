cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_r02h.log 2>&1; tail -1 gpurun_out/build_r02h.log
timeout 300 python tools/probe_physics.py run 2>&1 | tail -12
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/test_r02h.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/test_r02h.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
BENCH_WATCHDOG_S=150 timeout 400 python bench.py > gpurun_out/bench_r02h.json 2> gpurun_out/bench_r02h.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/bench_r02h.json; grep "bench " gpurun_out/bench_r02h.err | tail -4
timeout 900 bash tools/profile.sh r02h 2>&1 | tail -25
