"""Multi-GPU: environments shard embarrassingly, a contiguous env slice per GPU, no collectives on the data path.

Every kernel indexes by env and touches only that env's rows (reference: kernels.cu:185-209,298,330,412), so the 8 GPUs
of a node each own ``n_envs/8`` envs and never talk to each other; xGMI/RCCL carry nothing but the benchmark's barrier.
One process per GPU (the reference's implicit current-device model, common.h:39-41).
"""
import torch
from . import cuda


def env_slice(n_envs, rank, world_size, cost=None):
    """The contiguous ``[start, stop)`` slice of envs owned by ``rank``.

    Without ``cost`` sizes differ by at most one. With ``cost`` - a per-env weight, e.g. :func:`render_cost` - the cuts
    sit where the running total crosses ``rank/world_size`` of the whole, so every rank carries the same share of the
    work rather than of the envs (floorplans differ several-fold in wall count); every rank still gets at least one env."""
    if not (0 <= rank < world_size):
        raise ValueError(f'rank {rank} outside world of {world_size}')
    if cost is None:
        base, extra = divmod(n_envs, world_size)
        start = rank*base + min(rank, extra)
        return start, start + base + (1 if rank < extra else 0)
    cost = torch.as_tensor(cost, dtype=torch.float64).reshape(-1).cpu()
    if len(cost) != n_envs or n_envs < world_size or not bool((cost >= 0).all()):
        raise ValueError('cost must hold one non-negative weight per env, and there must be an env per rank')
    total = cost.cumsum(0)
    # cut r: after the first env at which the running total reaches r/world_size of the whole
    targets = total[-1]*torch.arange(1, world_size, dtype=torch.float64)/world_size
    cuts = torch.searchsorted(total, targets).clamp(max=n_envs - 1) + 1
    bounds = [0] + cuts.tolist() + [n_envs]
    for r in range(1, world_size + 1):                      # at least one env each, in order
        bounds[r] = min(max(bounds[r], bounds[r - 1] + 1), n_envs - (world_size - r))
    return bounds[rank], bounds[rank + 1]


def render_cost(scenery, res):
    """Per-env weight lines x agents x rays - what the raycast's work scales with."""
    return scenery.lines.widths.double()*scenery.n_agents*res


def shard_scenery(scenery, rank, world_size, device=None, cost=None):
    """The part of ``scenery`` that ``rank`` owns, as an independent :class:`~megastep_amd.cuda.Scenery` on ``device``.

    ``lights`` and ``lines`` are ragged per env and sliced by env; ``textures`` and ``baked`` are ragged per line and
    sliced by the env slice's line range (as ``Scenery.state`` does, reference: common.h:203-211). Baked lighting is
    carried over, so a shard never needs re-baking. ``cost``: see :func:`env_slice`."""
    start, stop = env_slice(len(scenery.lines), rank, world_size, cost)
    device = scenery.model.device if device is None else device
    if stop == start:
        raise ValueError(f'rank {rank} of {world_size} would own no envs out of {len(scenery.lines)}')
    l0, l1 = int(scenery.lines.starts[start]), int(scenery.lines.ends[stop - 1])
    move = lambda r: type(r)(r.vals.to(device).contiguous().clone(), r.widths.to(device).contiguous().clone())
    geom = None
    if getattr(scenery, 'geom', None) is not None:
        # representatives are re-elected inside the shard: the first env of the slice with the same floorplan
        parent_rep = scenery.geom[start:stop].long()
        _, group = torch.unique(parent_rep, return_inverse=True)
        here = torch.arange(stop - start, device=group.device)
        first = torch.full((int(group.max()) + 1,), stop - start, device=group.device).scatter_reduce_(0, group, here, 'amin')
        geom = first[group].to(torch.int32).to(device).contiguous()
    out = cuda.Scenery(
        n_agents=scenery.n_agents,
        lights=move(scenery.lights[start:stop]),
        lines=move(scenery.lines[start:stop]),
        textures=move(scenery.textures[l0:l1]),
        model=scenery.model.to(device).clone(), geom=geom)
    out.baked.vals.copy_(scenery.baked[l0:l1].vals)
    # the light grid (what ms_bake caches about which lights reach which cells) is per floorplan: carry the slice's over
    parent = getattr(scenery, '_lg', None)
    if parent is not None and parent[0] is not None and out.model.is_cuda:
        out._lg = _shard_light_grid(parent, start, stop, device, geom)
    # the wall grid (per floorplan, too) is rebuilt from the shard's own walls: two launches
    if getattr(scenery, '_wg', None) is not None and out.model.is_cuda:
        out._build_wall_grid()
    return out


def _shard_light_grid(lg, start, stop, device, geom=None):
    """Envs [start, stop) of a baked light grid (cuda.Scenery._light_grid's tuple): the cells of the slice's
    representative envs (`geom`, slice-local; None = every env its own) back to back, the candidate lists repacked into
    a pool of their own (the parent's pool is filled in no particular order)."""
    vals, starts, grid, cell, _, lists, pool = lg[:7]
    pool_rows = lg[7] if len(lg) > 7 else None
    dev = vals.device
    n = stop - start
    rep = torch.arange(n, device=dev) if geom is None else geom.long().to(dev)
    sub_geom = grid[start:stop]
    cells = (sub_geom[:, 2]*sub_geom[:, 3]).long()
    own = cells*(rep == torch.arange(n, device=dev))
    new_starts = (own.cumsum(0) - own)
    total = int(own.sum())
    # for every cell of the shard, the parent's row it copies
    src = torch.arange(total, device=dev) + torch.repeat_interleave(starts[start:stop].long() - new_starts, own, output_size=total)
    rows = lists[src].long() & 0xffffffff
    count = torch.where(rows[:, 1] != 0, rows[:, 1] & 0x7fffffff, torch.zeros_like(rows[:, 1]))
    first = count.cumsum(0) - count                                   # 0-based position in the new pool's payload
    cell_of = torch.repeat_interleave(torch.arange(len(count), device=dev), count)
    take = rows[cell_of, 0] + (torch.arange(int(count.sum()), device=dev) - first[cell_of])
    sub_pool = torch.cat([count.sum()[None].to(pool.dtype), pool[take]])
    sub_rows = None if pool_rows is None else torch.cat([torch.zeros_like(pool_rows[:1]), pool_rows[take]]).to(device).contiguous()
    sub_lists = torch.stack([torch.where(rows[:, 1] != 0, first + 1, torch.zeros_like(first)), rows[:, 1]], 1).to(torch.int32)
    pad = lambda t: torch.cat([t, torch.zeros_like(t[:1])])               # (the row rays outside the last env's grid read)
    return (pad(vals[src]).to(device).contiguous().clone(), new_starts[rep].to(torch.int32).to(device).contiguous(),
            sub_geom.to(device).contiguous().clone(), cell, max(int(cells.max()), 1), pad(sub_lists).to(device).contiguous(),
            sub_pool.to(device).contiguous(), sub_rows)


def max_over_ranks(seconds, device=None):
    """The slowest rank's time - what a synchronous multi-GPU step rate is limited by. A no-op without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
