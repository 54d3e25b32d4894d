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
def per_wave(t): 
    v = torch.where(dyn, t, torch.zeros_like(t)).max(-1).values
    return v[dyn.any(-1)]
tot, setup, rank, ph1 = per_wave(r.distances), per_wave(r.locations), per_wave(r.dots), per_wave(r.screen[...,0])
nd = dyn.sum(-1)[dyn.any(-1)].float()
rest = tot - setup - rank - ph1
order = tot.argsort()
for name, sel in [('median 10%', order[len(order)//2-100:len(order)//2+100]), ('slowest 1%', order[-24:]), ('slowest', order[-3:])]:
    print(name, 'total', tot[sel].mean().item(), 'setup', setup[sel].mean().item(), 'rank', rank[sel].mean().item(), 'phase1', ph1[sel].mean().item(), 'rest', rest[sel].mean().item(), 'rays', nd[sel].mean().item())
