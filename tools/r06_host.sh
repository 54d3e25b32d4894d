#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6f; O=gpurun_out/r6f
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_gpu_envs.py tests/test_gpu_parity.py tests/test_gpu_depth.py tests/test_gpu_config.py tests/test_gpu_step_render.py -q --tb=short -p no:cacheprovider -x > $O/test.log 2>&1; echo "pytest rc=$?"; tail -5 $O/test.log
python tools/env_host_profile.py 2>&1 | head -40 > $O/host_profile.txt; sed -n 2,2p $O/host_profile.txt
python tools/host_overhead.py 2>&1 | tail -8
