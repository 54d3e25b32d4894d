cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_r02b.log 2>&1
MEGASTEP_RENDER_IMPL=v2 timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/test_r02b_v2.log 2>&1; echo "v2 pytest rc=$?"; tail -15 gpurun_out/test_r02b_v2.log
timeout 300 python -m pytest tests/test_gpu_scale.py -m gpu -q --tb=short -p no:cacheprovider -k c3 > gpurun_out/test_r02b_c3.log 2>&1; echo "c3 rc=$?"; tail -3 gpurun_out/test_r02b_c3.log
timeout 600 bash tools/ab_impl.sh "pairs v2"
BENCH_WATCHDOG_S=100 timeout 420 python bench.py --steps 100 > gpurun_out/bench_r02b.json 2> gpurun_out/bench_r02b.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/bench_r02b.json; grep "bench " gpurun_out/bench_r02b.err | tail -12; grep -A12 "most recent call first" gpurun_out/bench_r02b.err | head -40
