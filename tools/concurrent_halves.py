"""Experiment: the headline step as ONE batch of 4096 envs against TWO batches of 2048 stepped side by side on two streams
(as parallel branches of one HIP graph) - do the halves fill each other's drains?   usage: python tools/concurrent_halves.py [parts]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from megastep_amd import cuda, modules

bench.PLAN_WORKERS = 0
dev = bench._Gpu(0)
K, W = 20, 5
PARTS = int(sys.argv[1]) if len(sys.argv) > 1 else 2


def prepared(n, seed):
    core, _ = bench.build_world(n, 4, 64, 130., dev.device, seed=seed, n_unique=max(n//4, 1))
    torch.manual_seed(seed)
    mover = modules.MomentumMovement(core)
    agents, scenery = core.agents, core.scenery
    actions = torch.randint(0, 7, (K + W, n, 4), device=dev.device)
    hot = dev.hot_path(scenery)
    start = (agents.angles.clone(), agents.positions.clone())
    vel0 = torch.empty((K + W, n, 4, 2), device=dev.device); ang0 = torch.empty((K + W, n, 4), device=dev.device)
    for i in range(K + W):
        delta = mover._actionset[actions[i]]
        agents.angvelocity[:] = (1 - mover.decay)*agents.angvelocity + delta.angvelocity
        agents.velocity[:] = (1 - mover.decay)*agents.velocity + modules.to_global_frame(agents.angles, delta.velocity)
        vel0[i], ang0[i] = agents.velocity, agents.angvelocity
        hot(agents)
    vel, ang = vel0.clone(), ang0.clone()
    views = [cuda.Agents(agents.angles, agents.positions, ang[i], vel[i]) for i in range(K + W)]

    def rewind():
        agents.angles.copy_(start[0]); agents.positions.copy_(start[1]); vel.copy_(vel0); ang.copy_(ang0)
        for i in range(W):
            hot(views[i])
    return dict(hot=hot, views=views, rewind=rewind)


def timed(replay, worlds, reps=40):
    ts = []
    for _ in range(reps):
        for w in worlds:
            w['rewind']()
        dev.sync(); t0 = time.perf_counter(); replay(); dev.sync()
        ts.append(time.perf_counter() - t0)
    return 1e3*np.median(ts)/K


whole = prepared(4096, 1)
g = dev.graph(lambda: [whole['hot'](whole['views'][W + i]) for i in range(K)]); g()
print(f'one batch of 4096 envs:                   {timed(g, [whole]):.4f} ms/step', flush=True)
del whole; torch.cuda.empty_cache()

parts = [prepared(4096//PARTS, 1 + p) for p in range(PARTS)]
streams = [torch.cuda.Stream() for _ in range(PARTS - 1)]


def free_running():
    main = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(main)
    for p, w in enumerate(parts):
        with torch.cuda.stream(streams[p - 1] if p else main):
            for i in range(K):
                w['hot'](w['views'][W + i])
    for s in streams:
        main.wait_stream(s)


def joined_each_step():
    main = torch.cuda.current_stream()
    for i in range(K):
        for s in streams:
            s.wait_stream(main)
        for p, w in enumerate(parts):
            with torch.cuda.stream(streams[p - 1] if p else main):
                w['hot'](w['views'][W + i])
        for s in streams:
            main.wait_stream(s)


def one_after_the_other():
    for i in range(K):
        for w in parts:
            w['hot'](w['views'][W + i])


for name, fn in (('one after the other', one_after_the_other), ('side by side, joined every step', joined_each_step), ('side by side, free-running', free_running)):
    g = dev.graph(fn); g()
    print(f'{PARTS} batches of {4096//PARTS}, {name:32s}: {timed(g, parts):.4f} ms/step (of all 4096 envs)', flush=True)
