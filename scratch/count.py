import sys; sys.path.insert(0,'.')
import torch, numpy as np
import bench
from megastep_amd import cuda, modules
core,_ = bench.build_world(4096, 4, 64, 130., torch.device('cuda'), seed=1)
mover = modules.MomentumMovement(core)
for i in range(30):
    class D: actions = torch.randint(0,7,(4096,4),device='cuda')
    mover(D)
r = cuda.render(core.scenery, core.agents)
it = r.locations[...,0]; mine = r.dots; hits = r.distances; inc = r.screen[...,0,0]
L = core.scenery.lines.widths.float()
print('lines/env', L.mean().item(), 'chunks', torch.ceil(L/64).mean().item())
print('iterations/wave', it.mean().item(), 'max', it.max().item())
print('lines kept by interval/clip per wave', inc.mean().item())
print('pairs per lane (own group bits)', mine.mean().item(), ' -> mean over groups; max-group per wave', mine.max(-1).values.mean().item())
print('hits per lane', hits.mean().item(), 'hit fraction of lane-pairs', (hits.sum()/mine.sum()).item())
