cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/test_r02z.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/test_r02z.log
for impl in pairs seq; do MEGASTEP_RENDER_IMPL=$impl timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1; done
bash tools/ab_variants.sh "prev:v2 main:v2 prev:v2 main:v2"
bash tools/ab_variants.sh "prev:v2 main:v2" --envs 32768 --agents 1 --res 256 --large --unique 64 --fast-build
bash tools/ab_variants.sh "prev:v2 main:v2" --envs 16384
bash tools/ab_variants.sh "prev:v2 main:v2" --res 128
