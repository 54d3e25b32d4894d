cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_r02i.log 2>&1; tail -1 gpurun_out/build_r02i.log
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/test_r02i.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/test_r02i.log
timeout 300 python tools/probe_physics.py run 2>&1 | tail -11
MEGASTEP_HIP_LIB=$PWD/megastep_amd/csrc/variants/preload.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "step_matches" 2>&1 | tail -2
timeout 600 bash tools/ab_variants.sh "main:v2 preload:v2"
