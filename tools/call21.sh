cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_r02u.log 2>&1; tail -1 gpurun_out/build_r02u.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_envs.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -2
MEGASTEP_PHYS_EPW=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_envs.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -2
for shape in "" "--envs 32768 --agents 1 --res 256 --large --unique 64 --fast-build"; do
  timeout 600 bash tools/ab_variants.sh "main:v2 phys_guarded:v2" $shape
  for e in 2 4; do MEGASTEP_PHYS_EPW=$e timeout 600 bash tools/ab_variants.sh "main:v2" $shape; done
done
