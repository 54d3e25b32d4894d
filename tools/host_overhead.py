import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from megastep_amd import cuda, core, scene, toys
np.random.seed(0)
c = core.Core(scene.scenery(4*[toys.box()], 2), res=64)
c.agents.positions[:] = 3.
for f, name in ((lambda: cuda.physics(c.scenery, c.agents), 'physics'), (lambda: cuda.render(c.scenery, c.agents), 'render')):
    for _ in range(200): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(2000): f()
    host = (time.perf_counter() - t)/2000
    torch.cuda.synchronize()
    print(f'{name}: {host*1e6:.1f} us of host time per call')
