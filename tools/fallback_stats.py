import sys; sys.path.insert(0,'.')
import torch, numpy as np, bench
from megastep_amd import cuda, modules
core,_ = bench.build_world(4096, 4, 64, 130., torch.device('cuda'), seed=1)
mover = modules.MomentumMovement(core)
class D: pass
out=[]
for i in range(120):
    D.actions = torch.randint(0,7,(4096,4),device='cuda'); mover(D)
    r = cuda.render(core.scenery, core.agents, telemetry=True)
    out.append(r._telemetry[:3].tolist())
print('per step [queued groups, rays in sequential fold, lane-parallel waves]:'); print(out[::4])
