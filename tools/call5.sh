cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_r02e.log 2>&1; tail -2 gpurun_out/build_r02e.log
timeout 300 python tools/probe_v2.py run 2>&1 | tail -16
timeout 300 python tools/probe_run.py 2>&1 | tail -14
timeout 400 bash tools/pmc_impl.sh "pairs v2"
MEGASTEP_PHYS_WPB=4 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "step_matches or more_than_64" 2>&1 | tail -2
for w in 1 4; do MEGASTEP_PHYS_WPB=$w timeout 300 bash tools/ab_variants.sh "main:pairs"; done
