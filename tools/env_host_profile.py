"""Where the HOST's share of an Explorer step goes (cProfile of env.step(), GPU work asynchronous)."""
import cProfile, pstats, sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from megastep_amd import cubicasa, arrdict
from megastep_amd.demo import Explorer
pool = cubicasa.sample(64, n_unique=64)
np.random.seed(0); torch.manual_seed(0)
e = Explorer(4096, geometries=[pool[i % 64] for i in range(4096)])
e.reset()
acts = torch.randint(0, 7, (400, 4096, 1), device='cuda')
for i in range(50): e.step(arrdict.arrdict(actions=acts[i]))
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(50, 350): e.step(arrdict.arrdict(actions=acts[i]))
host = (time.perf_counter() - t)/300
torch.cuda.synchronize(); total = (time.perf_counter() - t)/300
print(f'host {host*1e6:.1f} us per step, with the GPU {total*1e6:.1f} us')
pr = cProfile.Profile(); pr.enable()
for i in range(350, 400): e.step(arrdict.arrdict(actions=acts[i]))
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumtime').print_stats(28)
