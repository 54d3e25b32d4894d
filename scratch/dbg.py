import sys; sys.path.insert(0,'.')
import numpy as np, torch
from tests import util
from tests.test_gpu_parity import _world
from megastep_amd import cuda
c, geometries = _world(2, 2, 64, 70, toy='column')
ref = util.OracleWorld(c); ref.bake(); ref.pull_baked(c)
rng = np.random.RandomState(7)
for step in range(2):
    util.random_velocities(c, rng, speed=4. if step % 2 else 40.)
    ref.pull_agents(c)
    p = cuda.physics(c.scenery, c.agents); r = cuda.render(c.scenery, c.agents)
    ref.physics(); want = ref.render()
    idx = want['indices']
    for n,a,ray in np.argwhere((idx>=0)&(idx<16)):
        l0 = idx[n,a,ray]; loc = want['locations'][n,a,ray]
        ln = ref.scene.lines_vals[ref.scene.lines_starts[n]+l0]
        C = ln[0]*(1-loc)+ln[1]*loc
        lights = c.scenery.lights[n].cpu().numpy()
        walls = ref.scene.lines_vals[ref.scene.lines_starts[n]+16: ref.scene.lines_starts[n]+ref.scene.lines_widths[n]]
        acc = .1; flags=[]
        for I in lights:
            U = C - I[:2]; blocked=False
            for w in walls:
                V = w[1]-w[0]; d = U[0]*V[1]-U[1]*V[0]
                if abs(d) < 1e-3: continue
                PQ = w[0]-I[:2]; s_ = (PQ[0]*V[1]-PQ[1]*V[0])/d; t_ = (PQ[0]*U[1]-PQ[1]*U[0])/d
                if 0<t_<1 and 0<s_<.999: blocked=True
            flags.append(blocked)
            if not blocked: acc += 2*I[2]/max(((I[:2]-C)**2).sum(),1)
        print('step',step,(n,a,ray),'idx',l0,'C',C,'oracle flags',flags,'acc',acc,'device acc', r.dots[n,a,ray].item(), 'target pos', c.agents.positions[n, l0//8].cpu().numpy())
