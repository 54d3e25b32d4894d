import sys; sys.path.insert(0,'.')
import torch, numpy as np
import bench
from megastep_amd import cuda, modules
core,_ = bench.build_world(4096, 1, 64, 130., torch.device('cuda'), seed=1)   # A=1: no dynlight kernel overwriting
mover = modules.MomentumMovement(core)
for i in range(30):
    class D: actions = torch.randint(0,7,(4096,1),device='cuda')
    mover(D)
r = cuda.render(core.scenery, core.agents)
q = torch.tensor([.1,.5,.9,.99,1.0], device='cuda')
scan = r.locations[...,0]; fb = r.dots[...,0]; amb = r.distances[...,0]
it = (r.indices[...,0]//100000).float(); pairs = (r.indices[...,0]%100000).float()
print('main part cycles quantiles', torch.quantile(scan.flatten(), q).tolist())
print('fallback cycles quantiles', torch.quantile(fb.flatten(), q).tolist())
print('ambiguous rays per wave: mean', amb.mean().item(), 'frac waves with any', (amb>0).float().mean().item(), 'max', amb.max().item())
print('pair iterations per wave', it.mean().item(), 'pairs per wave', pairs.mean().item(), 'chunks', torch.ceil(core.scenery.lines.widths.float()/64).mean().item())
w = amb.flatten().argmax().item()
n = w // 1; print('env', n, 'amb', amb.flatten()[w].item(), 'pos', core.agents.positions[n,0].tolist(), 'ang', core.agents.angles[n,0].item())
import os
os.environ.pop('MEGASTEP_HIP_LIB', None)
