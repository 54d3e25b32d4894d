"""Multi-GPU: environments shard embarrassingly, a contiguous env slice per GPU, no collectives on the data path.

Every kernel indexes by env and touches only that env's rows (reference: kernels.cu:185-209,298,330,412), so the 8 GPUs
of a node each own ``n_envs/8`` envs and never talk to each other; xGMI/RCCL carry nothing but the benchmark's barrier.
One process per GPU (the reference's implicit current-device model, common.h:39-41).
"""
import torch
from . import cuda


def env_slice(n_envs, rank, world_size):
    """The contiguous ``[start, stop)`` slice of envs owned by ``rank``; sizes differ by at most one."""
    if not (0 <= rank < world_size):
        raise ValueError(f'rank {rank} outside world of {world_size}')
    base, extra = divmod(n_envs, world_size)
    start = rank*base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_scenery(scenery, rank, world_size, device=None):
    """The part of ``scenery`` that ``rank`` owns, as an independent :class:`~megastep_amd.cuda.Scenery` on ``device``.

    ``lights`` and ``lines`` are ragged per env and sliced by env; ``textures`` and ``baked`` are ragged per line and
    sliced by the env slice's line range (as ``Scenery.state`` does, reference: common.h:203-211). Baked lighting is
    carried over, so a shard never needs re-baking."""
    start, stop = env_slice(len(scenery.lines), rank, world_size)
    device = scenery.model.device if device is None else device
    if stop == start:
        raise ValueError(f'rank {rank} of {world_size} would own no envs out of {len(scenery.lines)}')
    l0, l1 = int(scenery.lines.starts[start]), int(scenery.lines.ends[stop - 1])
    move = lambda r: type(r)(r.vals.to(device).contiguous().clone(), r.widths.to(device).contiguous().clone())
    out = cuda.Scenery(
        n_agents=scenery.n_agents,
        lights=move(scenery.lights[start:stop]),
        lines=move(scenery.lines[start:stop]),
        textures=move(scenery.textures[l0:l1]),
        model=scenery.model.to(device).clone())
    out.baked.vals.copy_(scenery.baked[l0:l1].vals)
    # the light grid (what ms_bake caches about which lights reach which cells) is per env too: carry its slice over
    parent = getattr(scenery, '_lg', None)
    if parent is not None and parent[0] is not None and out.model.is_cuda:
        out._lg = _shard_light_grid(parent, start, stop, device)
    return out


def _shard_light_grid(lg, start, stop, device):
    """Envs [start, stop) of a baked light grid (cuda.Scenery._light_grid's tuple): the verdict rows as they are, the
    candidate lists repacked into a pool of their own (the parent's pool is filled in no particular order)."""
    vals, starts, geom, cell, _, lists, pool = lg
    c0 = int(starts[start])
    c1 = int(starts[stop]) if stop < len(starts) else vals.shape[0]
    sub_starts = (starts[start:stop] - c0).to(device).contiguous()
    sub_geom = geom[start:stop].to(device).contiguous().clone()
    cells = (sub_geom[:, 2]*sub_geom[:, 3]).long()
    rows = lists[c0:c1].long() & 0xffffffff
    count = torch.where(rows[:, 1] != 0, rows[:, 1] & 0x7fffffff, torch.zeros_like(rows[:, 1]))
    first = count.cumsum(0) - count                                   # 0-based position in the new pool's payload
    cell_of = torch.repeat_interleave(torch.arange(len(count), device=count.device), count)
    src = rows[cell_of, 0] + (torch.arange(int(count.sum()), device=count.device) - first[cell_of])
    sub_pool = torch.cat([count.sum()[None].to(pool.dtype), pool[src]])
    sub_lists = torch.stack([torch.where(rows[:, 1] != 0, first + 1, torch.zeros_like(first)), rows[:, 1]], 1).to(torch.int32)
    return (vals[c0:c1].to(device).contiguous().clone(), sub_starts.to(torch.int32), sub_geom, cell, int(cells.max()),
            sub_lists.to(device).contiguous(), sub_pool.to(device).contiguous())


def max_over_ranks(seconds, device=None):
    """The slowest rank's time - what a synchronous multi-GPU step rate is limited by. A no-op without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
