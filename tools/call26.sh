cd $GRAFT_REPO_ROOT
timeout 300 python tools/probe_slices.py 2>&1 | tail -12
timeout 300 python tools/probe_slices.py --res 128 2>&1 | tail -9
