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
print('dynamic rays', dyn.sum().item(), 'lights evaluated per dyn ray (wave-level count)', r.dots[dyn].mean().item(), 'saturated frac', r.locations[dyn].mean().item())
print('lights/env', core.scenery.lights.widths.float().mean().item())
print(torch.bincount(r.dots[dyn].long())[:30])
