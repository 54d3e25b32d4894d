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
idx = r.indices
ah = (idx>=0)&(idx<32)
print('frac rays hitting agents', ah.float().mean().item(), 'frac waves with agent hit', ah.any(-1).float().mean().item())
print('rays per hit-wave', ah.sum(-1)[ah.any(-1)].float().mean().item())
print('miss frac', (idx<0).float().mean().item())
print('progress<1 frac', (cuda.physics(core.scenery, core.agents).progress<1).float().mean().item())
