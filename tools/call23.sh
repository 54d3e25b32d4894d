cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_r02w.log 2>&1; tail -1 gpurun_out/build_r02w.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_envs.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -2
MEGASTEP_HIP_LIB=$PWD/megastep_amd/csrc/variants/phys_ldsbox.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -1
timeout 200 python tools/pair_stats.py v2 2>&1 | tail -1
for shape in "" "--envs 32768 --agents 1 --res 256 --large --unique 64 --fast-build"; do
  timeout 600 bash tools/ab_variants.sh "main:v2 phys_ldsbox:v2 main:v2 phys_ldsbox:v2" $shape
done
timeout 400 python bench.py --no-cpu-baseline --no-env-fps --steps 100 --envs 16384 2>/dev/null | cut -c1-200
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in libmegastep_hip.so variants/phys_ldsbox.so; do
MEGASTEP_HIP_LIB=$PWD/megastep_amd/csrc/$lib rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace -d gpurun_out/pmcP -o p --output-format csv -- python bench.py --no-cpu-baseline --no-env-fps --no-graph --steps 8 --warmup 2 > gpurun_out/pmcP.log 2>&1
python - <<PY
import pandas as pd
d = pd.read_csv('gpurun_out/pmcP/p_counter_collection.csv'); d = d[d.Kernel_Name.str.contains('physics_kernel')]
g = d.groupby('Counter_Name').Counter_Value.mean(); print('$lib', (g/g.SQ_WAVES).round(1).to_dict())
PY
done
