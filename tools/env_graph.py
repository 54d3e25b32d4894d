"""Whole env.step() of the demo envs captured in a HIP graph (torch.cuda.CUDAGraph) and replayed: what the step costs
once launch overhead is out of the picture. Needs every op of the step to be sync-free, which is itself the check."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from megastep_amd import cubicasa, arrdict
from megastep_amd.demo import Explorer, Deathmatch

pool = cubicasa.sample(256, n_unique=512)


def run(env, n, steps=100, warm=10):
    A = env.action_space.shape[0]
    env.reset()
    actions = torch.randint(0, 7, (n, A), device='cuda')
    decision = arrdict.arrdict(actions=actions)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm-up on a side stream, as graph capture asks for
        for _ in range(warm):
            env.step(decision)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        env.step(decision)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t)/steps
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        world = env.step(decision)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        actions.random_(0, 7)                          # new actions, written into the captured input
        g.replay()
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t)/steps
    assert torch.isfinite(world.obs.rgb).all()
    return eager, graphed


np.random.seed(0); torch.manual_seed(0)
e = Explorer(4096, geometries=[pool[i % 256] for i in range(4096)])
a, b = run(e, 4096)
print(f'Explorer(4096): eager {a*1e3:.3f} ms/step ({4096/a/1e6:.2f} M FPS), graph replay {b*1e3:.3f} ms/step ({4096/b/1e6:.2f} M FPS)')
del e; torch.cuda.empty_cache()
d = Deathmatch(16384, 4, geometries=[pool[i % 256] for i in range(4096)])
a, b = run(d, 16384)
print(f'Deathmatch(16384, 4): eager {a*1e3:.3f} ms/step ({16384/a/1e6:.2f} M FPS), graph replay {b*1e3:.3f} ms/step ({16384/b/1e6:.2f} M FPS)')
