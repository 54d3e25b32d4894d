#!/bin/bash
# Round 5, sixth GPU call: divisions without the range scaling where operands are in range (MS_DIV_INRANGE): parity, fuzz, A/B.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c6; O=gpurun_out/c6
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/test.log 2>&1; echo "pytest rc=$?"; tail -3 $O/test.log
timeout 900 python tools/fuzz_parity.py 6000 300 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
cp megastep_amd/csrc/libmegastep_hip.so megastep_amd/csrc/variants/product.so
bash tools/ab_libs.sh "variants/nodiv variants/product variants/nodiv variants/product" "" "--res 128 --fov 70" "--res 512 --fov 70" "--envs 32768 --agents 1 --res 256 --large --unique 64 --fast-build" "--agents 1" 2>&1 | tee $O/ab_div.txt
