"""render_kernel's NG (64-ray groups per wave) A/B: the bench protocol (K = 20, W = 5, HIP-graph replays + HIP events around
every render) per shape with the groups pinned through ms_debug_ray_groups.
usage: python tools/ab_groups.py [--depth-only] [--tail=ROUNDS ...] [--groups=1,4] [shape ...]     (--tail: the share of one-group
waves at the end of a launch of wide ones, ms_debug_ray_group_tail; several: each in turn)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from megastep_amd import _lib

SHAPES = {'c3': dict(n=4096, a=4, res=128, fov=70., u=1024), 'r256': dict(n=4096, a=4, res=256, fov=70., u=1024),
          'r512': dict(n=4096, a=4, res=512, fov=70., u=1024), 'r512x2': dict(n=8192, a=4, res=512, fov=70., u=1024),
          'explorer256x8': dict(n=32768, a=1, res=256, fov=130., u=1024, fast=True),
          'r512half': dict(n=2048, a=4, res=512, fov=70., u=1024), 'explorer256': dict(n=4096, a=1, res=256, fov=130., u=1024),
          'c5': dict(n=32768, a=1, res=256, fov=130., u=64, large=True, fast=True)}
dev = bench._Gpu(0)
h = _lib.lib()
DEPTH = '--depth-only' in sys.argv
TAILS = [float(a.split('=')[1]) for a in sys.argv[1:] if a.startswith('--tail=')] or [-1.]
GROUPS = [int(x) for a in sys.argv[1:] if a.startswith('--groups=') for x in a.split('=')[1].split(',')] or [1, 2, 4]
for name in [a for a in sys.argv[1:] if not a.startswith('--')] or list(SHAPES):
    sh = SHAPES[name]
    core, _ = bench.build_world(sh['n'], sh['a'], sh['res'], sh['fov'], dev.device, seed=1, n_unique=sh['u'], large=sh.get('large', False),
                                fast=sh.get('fast', False))
    start = (core.agents.angles.clone(), core.agents.positions.clone())
    for g, tail in [(g, t) for g in GROUPS for t in (TAILS if g > 1 else TAILS[:1])]:
        if 64*g > 2*sh['res']:
            continue
        h.ms_debug_ray_groups(g)
        h.ms_debug_ray_group_tail(tail, -1)
        core.agents.angles.copy_(start[0]); core.agents.positions.copy_(start[1])
        core.agents.velocity.zero_(); core.agents.angvelocity.zero_()
        m = bench.time_hot_path(dev, core, 20, 5, fields=('distances',) if DEPTH else None, eager_floor=.05, graph_floor=.15)
        print(f'{name:12s}{" depth-only" if DEPTH else ""} groups {g}{"" if g == 1 else f" tail {tail:g}"}: {1e3*np.median(m["runs"])/20:.4f} ms/step, render {1e3*np.median(m["render_each"]):.1f} us (HIP events, eager)', flush=True)
    h.ms_debug_ray_groups(0)
    del core
    torch.cuda.empty_cache()
