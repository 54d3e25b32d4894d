#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c25; O=gpurun_out/c25
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 600 python -m pytest tests/test_gpu_numerics.py tests/test_abi.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -15
