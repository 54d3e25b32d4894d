import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from megastep_amd import cubicasa, arrdict
from megastep_amd.demo import Explorer, Deathmatch
pool = cubicasa.sample(256, n_unique=512)
def run(env, n, steps=60, warm=10):
    A = env.action_space.shape[0]
    env.reset()
    acts = torch.randint(0, 7, (steps+warm, n, A), device='cuda')
    for i in range(warm): env.step(arrdict.arrdict(actions=acts[i]))
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(steps): env.step(arrdict.arrdict(actions=acts[warm+i]))
    torch.cuda.synchronize(); dt = (time.perf_counter()-t)/steps
    return dt
np.random.seed(0); torch.manual_seed(0)
e = Explorer(4096, geometries=[pool[i%256] for i in range(4096)])
dt = run(e, 4096); print(f'Explorer(4096) res=256->64px: {dt*1e3:.3f} ms/step, {4096/dt/1e6:.2f} M FPS (reference: 0.18 M on a 2080 Ti)')
del e; torch.cuda.empty_cache()
d = Deathmatch(16384, 4, geometries=[pool[i%256] for i in range(4096)])
dt = run(d, 16384); print(f'Deathmatch(16384, 4) res=512->128px: {dt*1e3:.3f} ms/step, {16384/dt/1e6:.2f} M FPS (reference: 1.2 M on a 2080 Ti)')
