"""The wall grid's build under the library in use (MEGASTEP_HIP_LIB selects a variant, e.g. one built with -DMS_WG_SECTORS=0): for a
few worlds the seconds the grid took and a checksum of everything it holds - cell headers, vis entries, near rows - so that two
builds can be held against each other bit for bit.   usage: python tools/ab_wallgrid_scan.py [--c5]"""
import hashlib
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

bench.PLAN_WORKERS = 32


def digest(sc):
    h = hashlib.sha256()
    cells, starts, geom, cell, reach_lo, reach, near, pool, rows, pool_base = sc._wg
    for t in (cells, starts, geom, pool_base):
        h.update(t.cpu().numpy().tobytes())
    for t in (pool, rows):                                                # (large: a device-side fold first)
        v = t.reshape(-1).view(torch.int32).long()
        w = (torch.arange(v.numel(), device=v.device) % 1000003) + 1
        h.update(str(int((v*w).sum())).encode())
    return h.hexdigest()[:16]


worlds = [('headline 1024 plans', dict(n_envs=4096, n_agents=4, res=64, fov=130., n_unique=1024)),
          ('headline oblique', dict(n_envs=4096, n_agents=4, res=64, fov=130., n_unique=1024, oblique=True)),
          ('C2 4096 plans', dict(n_envs=4096, n_agents=1, res=64, fov=130., n_unique=4096)),
          ('large 256 plans', dict(n_envs=2048, n_agents=1, res=256, fov=130., n_unique=256, large=True, fast=True))]
if '--c5' in sys.argv:
    worlds.append(('C5 share 4096 large plans', dict(n_envs=32768, n_agents=1, res=256, fov=130., n_unique=4096, large=True, fast=True)))
for name, kw in worlds:
    core, _ = bench.build_world(device=torch.device('cuda'), seed=1, **kw)
    rep = core.scenery.grid_report()
    print(f"{name:28s} wall grid {rep['bake_seconds']['wall_grid']:7.3f} s  {rep['wall_grid']['bytes']/2**30:6.2f} GiB  "
          f"vis entries {rep['wall_grid']['vis_entries']:>11d}  digest {digest(core.scenery)}", flush=True)
    del core
    torch.cuda.empty_cache()
