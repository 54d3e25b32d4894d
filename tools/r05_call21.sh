#!/bin/bash
# Round 5, twenty-first GPU call: the N > 1 bench path on real HIP - two ranks sharing the box's one GPU (--share-gpu: plumbing, not rates).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c21; O=gpurun_out/c21
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 600 python bench.py --gpus 2 --share-gpu --steps 20 --warmup 5 > $O/bench_2ranks.json 2> $O/bench_2ranks.err; echo "2 ranks rc=$?"; tail -3 $O/bench_2ranks.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --share-gpu --steps 20 --warmup 5 > $O/bench_2ranks_torchrun.json 2> $O/bench_2ranks_torchrun.err; echo "torchrun 2 ranks rc=$?"
python - <<'PY'
import json
for f in ('bench_2ranks', 'bench_2ranks_torchrun'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/c21/{f}.json') if l.startswith('{')][-1])
        print(f, 'n_gpus', d['n_gpus'], 'value', round(d['value']/1e6, 1), 'ms/step', round(d['ms_per_step'], 5), 'per_rank', d['per_rank']['envs'], [round(x, 5) for x in d['per_rank']['ms_per_step']], 'regions', d['timed_regions']['count'], d['data'][:40])
    except Exception as e:
        print(f, 'no line', e)
PY
