import sys, time; sys.path.insert(0,'.')
import numpy as np, torch, bench
from megastep_amd import cuda, modules
t=time.time()
core,_ = bench.build_world(int(sys.argv[1]), 1, 256, 130., torch.device('cuda'), seed=1, n_unique=64, large=True)
torch.cuda.synchronize(); print('build s', time.time()-t, 'lines', core.scenery.lines.vals.shape, 'texels', core.scenery.textures.vals.shape, 'mem GB', torch.cuda.memory_allocated()/1e9)
N=core.n_envs
mover = modules.MomentumMovement(core)
class D: pass
for i in range(10):
    D.actions = torch.randint(0,7,(N,1),device='cuda'); mover(D); cuda.render(core.scenery, core.agents)
torch.cuda.synchronize(); t=time.perf_counter()
for i in range(30):
    cuda.physics(core.scenery, core.agents); r = cuda.render(core.scenery, core.agents)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/30
rb, pb = bench.algorithmic_bytes(core)
print(f'C5-like: {N} envs x 1 agent x 256 rays, {core.scenery.lines.vals.shape[0]/N:.0f} lines/env: {dt*1e3:.3f} ms/step, {N/dt/1e6:.2f} M env-steps/s, step algorithmic {(rb+pb)/1e6:.0f} MB -> {(rb+pb)/dt/1e9:.0f} GB/s')
