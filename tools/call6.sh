cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_r02f.log 2>&1; tail -1 gpurun_out/build_r02f.log
timeout 300 python tools/probe_v2.py run 2>&1 | tail -20
timeout 300 python tools/probe_run.py 2>&1 | tail -14
timeout 600 bash tools/ab_variants.sh "main:pairs w8:pairs main:v2 w8:v2"
