"""Does splitting the env batch into slices on separate HIP streams - so one slice's physics (latency-bound) runs under
another slice's render (VALU-bound) - shorten the physics+render step?  Every variant joins all streams at the end of
every step unless it says `free`.  usage: python tools/probe_slices.py [bench shape args]"""
import sys, time, argparse
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from megastep_amd import cuda, modules, sharding

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=4096); ap.add_argument('--agents', type=int, default=4)
ap.add_argument('--res', type=int, default=64); ap.add_argument('--steps', type=int, default=100)
ap.add_argument('--large', action='store_true'); ap.add_argument('--unique', type=int, default=512)
ap.add_argument('--fast-build', action='store_true')
args = ap.parse_args()
dev = torch.device('cuda', 0)
core, _ = bench.build_world(args.envs, args.agents, args.res, 130., dev, seed=1, n_unique=args.unique, large=args.large, fast=args.fast_build)
N, A, K = core.n_envs, core.n_agents, args.steps
scenery, agents = core.scenery, core.agents
torch.manual_seed(0)
vel = .5*torch.randn((K, N, A, 2), device=dev)
angvel = 30*torch.randn((K, N, A), device=dev)
cost = sharding.render_cost(scenery, args.res)

def world(S):
    shards = []
    for k in range(S):
        s, e = sharding.env_slice(N, k, S, cost)
        sc = scenery if S == 1 else sharding.shard_scenery(scenery, k, S, cost=cost)
        views = [cuda.Agents(agents.angles[s:e], agents.positions[s:e], angvel[i][s:e], vel[i][s:e]) for i in range(K)]
        shards.append((sc, views, {}))
    return shards

def phys(sh, i):
    sc, views, st = sh
    st['p'] = cuda.physics(sc, views[i], out=st.get('p'))

def rend(sh, i):
    sc, views, st = sh
    st['r'] = cuda.render(sc, views[i], out=st.get('r'))

def capture(S, mode):
    shards = world(S)
    for sh in shards:                    # warm (allocations outside the capture)
        phys(sh, 0); rend(sh, 0)
    torch.cuda.synchronize()
    side = [torch.cuda.Stream() for _ in range(S - 1)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        streams = [main] + side
        if mode == 'free':
            for st in side: st.wait_stream(main)
            for k, sh in enumerate(shards):
                with torch.cuda.stream(streams[k]):
                    for i in range(K): phys(sh, i); rend(sh, i)
            for st in side: main.wait_stream(st)
        else:
            for i in range(K):
                if mode == 'together':
                    for st in side: st.wait_stream(main)
                    for k, sh in enumerate(shards):
                        with torch.cuda.stream(streams[k]): phys(sh, i); rend(sh, i)
                elif mode == 'staggered':        # slice k's physics starts when slice k-1's has finished
                    prev = None
                    for k, sh in enumerate(shards):
                        with torch.cuda.stream(streams[k]):
                            if k: streams[k].wait_event(prev)
                            phys(sh, i)
                            prev = torch.cuda.Event(); prev.record(streams[k])
                            rend(sh, i)
                elif mode == 'phys-first':       # all physics on main one after another, renders fan out
                    evs = []
                    for k, sh in enumerate(shards):
                        phys(sh, i); e = torch.cuda.Event(); e.record(main); evs.append(e)
                    for k, sh in enumerate(shards):
                        with torch.cuda.stream(streams[k]):
                            if k: streams[k].wait_event(evs[k])
                            rend(sh, i)
                for st in side: main.wait_stream(st)
    return g, shards

for S, mode in [(1, 'together'), (2, 'together'), (2, 'staggered'), (4, 'staggered'), (2, 'phys-first'), (8, 'staggered'), (2, 'free'), (4, 'free')]:
    try:
        g, keep = capture(S, mode)
        g.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0)/K)
        print(f'slices {S} {mode:10s}: {1e6*min(ts):7.1f} us/step  ({N/min(ts)/1e6:.1f} M env-steps/s)', flush=True)
    except Exception as ex:
        print(f'slices {S} {mode}: failed: {type(ex).__name__}: {str(ex)[:200]}', flush=True)
    del g, keep
