import sys; sys.path.insert(0,'.')
import torch, numpy as np
import bench
from megastep_amd import cuda, modules
core,_ = bench.build_world(4096, 1, 64, 130., torch.device('cuda'), seed=1)
mover = modules.MomentumMovement(core)
for i in range(30):
    class D: actions = torch.randint(0,7,(4096,1),device='cuda')
    mover(D)
r = cuda.render(core.scenery, core.agents)
q = torch.tensor([.1,.5,.9,.99,.999,1.0], device='cuda')
it = (r.indices[...,0]//100000).float().flatten(); pairs = (r.indices[...,0]%100000).float().flatten()
print('pair iterations quantiles', torch.quantile(it, q).tolist(), 'mean', it.mean().item())
print('pairs quantiles', torch.quantile(pairs, q).tolist(), 'mean', pairs.mean().item())
scan = r.locations[...,0].flatten()
print('main cycles quantiles', torch.quantile(scan, q).tolist())
