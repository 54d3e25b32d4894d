#!/bin/bash
# Round 5, fourteenth GPU call: the pooled-observation instantiation against the plain one at 512 rays (Deathmatch's render).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c14; O=gpurun_out/c14
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 600 python tools/ab_envstep.py --res 512 --fov 70 --sub 4 --centre 2> $O/err.txt | tee $O/ab_obs512.txt
timeout 600 python tools/ab_envstep.py --agents 1 --res 256 --sub 4 2>> $O/err.txt | tee $O/ab_obs256.txt
