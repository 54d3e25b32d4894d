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
t = r.distances[dyn]
wave_t = torch.where(dyn, r.distances, torch.zeros_like(r.distances)).max(-1).values
wave_t = wave_t[wave_t>0]
q = torch.tensor([.1,.5,.9,.99,1.0], device='cuda')
print('active waves', wave_t.numel(), 'cycle quantiles', torch.quantile(wave_t, q).tolist(), 'mean', wave_t.mean().item())
nd = dyn.sum(-1)[dyn.any(-1)].float()
print('dyn rays per active wave quantiles', torch.quantile(nd, q).tolist())
# correlation
big = wave_t > torch.quantile(wave_t, .99)
print('rays in slowest 1% waves', nd[big].mean().item(), ' L of those envs', )
for name, t in [('setup(loads)', r.locations), ('rank+sync', r.dots), ('phase1', r.screen[...,0]), ('total', r.distances)]:
    v = t[dyn]
    print(name, 'median', v.median().item(), 'p90', torch.quantile(v, .9).item())
