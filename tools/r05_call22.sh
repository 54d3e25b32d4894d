#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c22; O=gpurun_out/c22
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 600 python tools/env_host_profile.py > $O/host_profile.txt 2>&1; tail -45 $O/host_profile.txt | cut -c1-150
