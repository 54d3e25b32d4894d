cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/test_r02z.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/test_r02z.log
bash tools/ab_variants.sh "prev:v2 main:v2 prev:v2 main:v2"
bash tools/ab_variants.sh "prev:v2 main:v2" --envs 32768 --agents 1 --res 256 --large --unique 64 --fast-build
bash tools/ab_variants.sh "prev:v2 main:v2" --envs 16384
bash tools/ab_variants.sh "prev:v2 main:v2" --envs 1024 --agents 16
timeout 300 python tools/probe_physics.py run 2>&1 | tail -11
