#!/bin/bash
# Round 5, fifteenth GPU call: pooled observations with shifts instead of run-time divisions, physics with a host-known divisor:
# the suite, a fuzz, and the same A/B as call 14.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c15; O=gpurun_out/c15
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/test.log 2>&1; echo "pytest rc=$?"; tail -3 $O/test.log
timeout 900 python tools/fuzz_parity.py 12000 300 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
MEGASTEP_PHYSICS_PACK=3 timeout 600 python tools/fuzz_parity.py 12300 100 > $O/fuzz_p3.log 2>&1; echo "fuzz pack 3 rc=$?"; tail -1 $O/fuzz_p3.log
timeout 600 python tools/ab_envstep.py --res 512 --fov 70 --sub 4 --centre 2> $O/err.txt | tee $O/ab_obs512.txt
timeout 600 python tools/ab_envstep.py --agents 1 --res 256 --sub 4 2>> $O/err.txt | tee $O/ab_obs256.txt
timeout 600 python tools/ab_envstep.py 2>> $O/err.txt | tee $O/ab_obs64.txt
