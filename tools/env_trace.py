"""Which kernels a Deathmatch / Explorer env.step() launches, per step (run under rocprofv3 --kernel-trace --stats):
python tools/env_trace.py deathmatch|explorer"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from megastep_amd import cubicasa, arrdict
from megastep_amd.demo import Explorer, Deathmatch
which = sys.argv[1] if len(sys.argv) > 1 else 'deathmatch'
pool = cubicasa.sample(256, n_unique=512)
geoms = [pool[i % 256] for i in range(4096)]
np.random.seed(0); torch.manual_seed(0)
env, n = (Deathmatch(16384, 4, geometries=geoms), 16384) if which == 'deathmatch' else (Explorer(4096, geometries=geoms), 4096)
env.reset()
acts = torch.randint(0, 7, (110, n, env.action_space.shape[0]), device='cuda')
for i in range(10):
    env.step(arrdict.arrdict(actions=acts[i]))
torch.cuda.synchronize()
print('MARK steps=100')
for i in range(100):
    env.step(arrdict.arrdict(actions=acts[10 + i]))
torch.cuda.synchronize()
