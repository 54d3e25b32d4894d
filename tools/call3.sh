cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_r02c.log 2>&1; tail -2 gpurun_out/build_r02c.log
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/test_r02c.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/test_r02c.log
timeout 200 python tools/pair_stats.py pairs v2 2>&1 | tail -3
timeout 700 bash tools/ab_variants.sh "v1:pairs v1_noclip:pairs v1_flag:pairs v1_noclip_flag:pairs main:v2 v2_eager:v2 v2_clip:v2 v2_noslp:v2 v2_noslp:pairs"
timeout 400 bash tools/pmc_impl.sh "pairs v2"
BENCH_WATCHDOG_S=150 timeout 400 python bench.py --steps 100 > gpurun_out/bench_r02c.json 2> gpurun_out/bench_r02c.err; echo "bench rc=$?"; cut -c1-3000 gpurun_out/bench_r02c.json; grep "bench " gpurun_out/bench_r02c.err | tail -14; grep -A8 "most recent call first" gpurun_out/bench_r02c.err | head -20
