cd $GRAFT_REPO_ROOT
timeout 300 python tools/probe_physics.py run 2>&1 | tail -12
