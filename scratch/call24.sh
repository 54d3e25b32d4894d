cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_r02f.log 2>&1; tail -1 gpurun_out/build_r02f.log
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/test_r02f.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/test_r02f.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for impl in pairs seq; do MEGASTEP_RENDER_IMPL=$impl timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1; done
timeout 400 python bench.py > gpurun_out/bench_r02f.json 2> gpurun_out/bench_r02f.err; echo "bench rc=$?"; cut -c1-330 gpurun_out/bench_r02f.json
rm -f gpurun_out/shapes_r02f.jsonl
for shape in "--envs 16384" "--envs 4096 --agents 1 --res 64" "--res 128" "--res 512" "--envs 32768 --agents 1 --res 256 --large --unique 64 --fast-build"; do
  timeout 400 python bench.py --no-cpu-baseline --no-env-fps --steps 100 $shape 2>/dev/null >> gpurun_out/shapes_r02f.jsonl
done
python - <<PY
import json
for l in open('gpurun_out/shapes_r02f.jsonl'):
    d = json.loads(l); print(d['config']['workload'][:60], '| graph ms %.4f  %.1f M env-steps/s | eager ms %.4f | render launch %.4f ms frac %.3f' % (d['ms_per_step'], d['value']/1e6, d['eager']['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))
PY
timeout 900 bash tools/profile.sh r02f 2>&1 | tail -30
