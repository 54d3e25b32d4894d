cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_r02r.log 2>&1; tail -1 gpurun_out/build_r02r.log
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/test_r02r.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/test_r02r.log
timeout 600 bash tools/ab_variants.sh "main:v2"
timeout 600 bash tools/ab_variants.sh "main:v2" --res 128
timeout 600 bash tools/ab_variants.sh "main:v2" --envs 32768 --agents 1 --res 256 --large --unique 64 --fast-build
timeout 300 python tools/probe_v2.py run 2>&1 | tail -13
timeout 300 python tools/probe_physics.py run 2>&1 | tail -10
