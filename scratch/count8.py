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
dyn = (r.indices>=0)&(r.indices<32)
nl = r.distances[dyn]; sat = r.locations[dyn]; unk = r.dots[dyn]
print('dyn rays', dyn.sum().item(), 'saturated frac', sat.mean().item(), 'mean unknown lights per ray', unk.mean().item(), 'rays with 0 unknown', (unk==0).float().mean().item())
fan_need = torch.where(dyn, r.distances, torch.zeros_like(r.distances)).max(-1).values[dyn.any(-1)]
print('active fans', fan_need.numel(), 'fans needing a sweep', (fan_need>0).float().mean().item(), 'lights swept per such fan', fan_need[fan_need>0].mean().item())
lg = core.scenery._lg[0]
w = lg.view(-1).long() & 0xffffffff
lit = sum(((w >> (2*k)) & 3 == 1).sum().item() for k in range(16)); dark = sum(((w >> (2*k)) & 3 == 2).sum().item() for k in range(16))
print('grid: lit', lit, 'dark', dark, 'cells', lg.shape[0], 'lights/env ~17 -> unknown frac', 1 - (lit+dark)/(lg.shape[0]*16.95))
