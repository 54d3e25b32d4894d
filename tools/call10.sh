cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_r02j.log 2>&1; tail -1 gpurun_out/build_r02j.log
timeout 300 python tools/probe_physics.py run 2>&1 | tail -11
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_envs.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -3
timeout 600 bash tools/ab_variants.sh "main:v2"
timeout 600 bash tools/ab_variants.sh "main:v2" --envs 32768 --agents 1 --res 256 --large --unique 64 --fast-build
