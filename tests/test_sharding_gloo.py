"""N>1 path on CPU: two gloo processes each take their contiguous env slice of one scenery; no data-path collective,
only the benchmark's barrier + max-over-ranks timing."""
import os
import socket
import sys
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from megastep_amd import cubicasa, scene, sharding, core
    n_envs = 5
    # every rank derives the same global scenery from the same seeds, then keeps only its slice
    np.random.seed(0)
    full = scene.scenery(cubicasa.sample(n_envs, n_unique=16), 2, device='cpu', random=np.random.RandomState(0), bake=False)
    mine = sharding.shard_scenery(full, rank, world)
    start, stop = sharding.env_slice(n_envs, rank, world)
    c = core.Core(mine, res=16)
    assert c.n_envs == stop - start
    # the slices tile the envs: gather only sizes/checksums (test-side bookkeeping, not a data-path collective)
    stats = torch.tensor([c.n_envs, mine.lines.vals.shape[0], mine.textures.vals.shape[0]], dtype=torch.int64)
    gathered = [torch.zeros_like(stats) for _ in range(world)]
    dist.all_gather(gathered, stats)
    total = torch.stack(gathered).sum(0)
    assert total.tolist() == [n_envs, full.lines.vals.shape[0], full.textures.vals.shape[0]]
    assert torch.equal(mine.lines.vals, full.lines[start:stop].vals)
    # timing reduction used by bench.py: the slowest rank sets the step rate
    dist.barrier()
    t = sharding.max_over_ranks(1.0 + rank)
    assert t == float(world)
    value = (n_envs*10)/t                      # whole-job units / slowest rank's time
    torch.save({'rank': rank, 'value': value, 'slice': (start, stop)}, os.path.join(out_dir, f'r{rank}.pt'))
    dist.destroy_process_group()


def test_two_rank_env_sharding(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path/f'r{r}.pt') for r in range(world)]
    assert [o['slice'] for o in outs] == [(0, 3), (3, 5)]
    assert outs[0]['value'] == outs[1]['value'] == 25.0


def test_max_over_ranks_without_a_process_group():
    sys.path.insert(0, ROOT)
    from megastep_amd import sharding
    assert sharding.max_over_ranks(0.25) == 0.25
