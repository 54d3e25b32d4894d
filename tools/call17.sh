cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_r02q.log 2>&1; tail -1 gpurun_out/build_r02q.log
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/test_r02q.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/test_r02q.log
for i in pairs seq; do MEGASTEP_RENDER_IMPL=$i timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1; done
timeout 600 bash tools/ab_variants.sh "main:v2 main:pairs"
timeout 600 bash tools/ab_variants.sh "main:v2" --res 128
timeout 600 bash tools/ab_variants.sh "main:v2" --envs 32768 --agents 1 --res 256 --large --unique 64 --fast-build
timeout 300 python tools/probe_v2.py run 2>&1 | tail -13
