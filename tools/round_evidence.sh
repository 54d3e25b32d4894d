#!/bin/bash
# A round's evidence in one gpurun call (round 6's, as it was run): the whole GPU suite, 300 fuzz seeds (every third on oblique plans), the bench lines (defaults; the
# driver's K = 20; 8 ranks sharing the one GPU), the per-shape rocprofv3 profiles (headline, C2 depth-only as two launches and as one).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6z; O=gpurun_out/r6z
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.log 2>&1; echo "build+smoke rc=$?"; tail -1 $O/build.log
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/test.log 2>&1; echo "pytest rc=$?"; tail -4 $O/test.log
timeout 900 python tools/fuzz_parity.py 1000 300 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line_steps20.json 2> $O/bench_line_steps20.err; echo "bench20 rc=$?"; cut -c1-200 $O/bench_line_steps20.json
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?"; cut -c1-200 $O/bench_line.json
timeout 600 python bench.py --gpus 8 --share-gpu --steps 20 --warmup 5 --no-cpu-baseline --no-env-fps --no-shapes 2> $O/bench_8ranks.err | grep '^{' > $O/bench_line_eight_ranks_sharing_one_gpu.json; echo "share rc=$?"; cut -c1-200 $O/bench_line_eight_ranks_sharing_one_gpu.json
for spec in "headline:" "c2d:--agents 1 --depth-only" "c2d_one_launch:--agents 1 --depth-only --one-launch" "c2_one_launch:--agents 1 --one-launch"; do
  tag=${spec%%:*}; shape=${spec#*:}
  timeout 1200 bash tools/profile.sh $tag $shape > $O/profile_$tag.log 2>&1; echo "profile $tag rc=$?"
  grep -E "render_kernel|physics_kernel" $O/profile_$tag.log | head -4 | cut -c1-200
  mkdir -p $O/prof; for f in kernel_stats.csv traffic.json sq_counters_mean_per_launch.csv valu_busy.csv; do cp gpurun_out/prof_$tag/$f $O/prof/${tag}_$f 2>/dev/null; done
  tail -1 gpurun_out/prof_$tag/bench_stats.log > $O/prof/${tag}_bench_line_under_rocprof.json
done
