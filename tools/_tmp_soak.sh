#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6m; O=gpurun_out/r6m
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 1500 python tools/fuzz_parity.py 4000 3000 > $O/fuzz_default.log 2>&1; echo "fuzz default rc=$?"; tail -1 $O/fuzz_default.log
MEGASTEP_RAY_GROUPS=4 timeout 600 python tools/fuzz_parity.py 7000 300 > $O/fuzz_ng4.log 2>&1; echo "fuzz ng4 rc=$?"; tail -1 $O/fuzz_ng4.log
MEGASTEP_RAY_GROUPS=2 MEGASTEP_RAY_GROUP_TAIL_ENVS=1 timeout 600 python tools/fuzz_parity.py 7300 200 > $O/fuzz_ng2.log 2>&1; echo "fuzz ng2 rc=$?"; tail -1 $O/fuzz_ng2.log
MEGASTEP_PHYSICS_PACK=3 timeout 600 python tools/fuzz_parity.py 7500 300 > $O/fuzz_pack3.log 2>&1; echo "fuzz pack3 rc=$?"; tail -1 $O/fuzz_pack3.log
