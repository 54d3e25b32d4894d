cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_r02g.log 2>&1; tail -1 gpurun_out/build_r02g.log
MEGASTEP_RENDER_IMPL=v2 timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/test_r02g_v2.log 2>&1; echo "v2 pytest rc=$?"; tail -8 gpurun_out/test_r02g_v2.log
timeout 200 python tools/pair_stats.py pairs v2 2>&1 | tail -2
timeout 300 python tools/probe_v2.py run 2>&1 | tail -18
timeout 600 bash tools/ab_variants.sh "main:pairs main:v2 v2_eager:v2"
timeout 400 bash tools/ab_variants.sh "main:pairs main:v2" --res 128
timeout 600 bash tools/ab_variants.sh "main:pairs main:v2" --envs 32768 --agents 1 --res 256 --large --unique 64 --fast-build
