"""Where a whole env.step() at the headline shape spends its time against the kernels-only step: the same world, the same box,
the same protocol (K steps as one HIP graph, replayed; HIP events around the graph), with the step's two launches swapped one
at a time from the plain instantiations to the ones an env uses:

    physics  plain                      | + movement prologue  | + movement + IMU (what MomentumMovement(imu=) launches)
    render   five planes (the bench's)  | pooled RGB + depth written by the kernel, no planes (what modules.render(observers=) asks for)

usage: python tools/ab_envstep.py [--steps 20] [--envs 4096]"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from megastep_amd import cuda, modules

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--envs', type=int, default=4096)
ap.add_argument('--agents', type=int, default=4)
ap.add_argument('--res', type=int, default=64)
ap.add_argument('--fov', type=float, default=130.)
ap.add_argument('--sub', type=int, default=1, help='rays per observation pixel of the pooled variants')
ap.add_argument('--centre', action='store_true', help='pooled variants also write the crosshair ids')
args = ap.parse_args()
bench.PLAN_WORKERS = 0
dev = bench._Gpu(0)
core, _ = bench.build_world(args.envs, args.agents, args.res, args.fov, dev.device, seed=1, n_unique=bench.plan_count(args.envs, args.agents))
N, A, K = core.n_envs, core.n_agents, args.steps
sc, ag = core.scenery, core.agents
mover, imu = modules.MomentumMovement(core), modules.IMU(core)
torch.manual_seed(0)
acts = torch.randint(0, 7, (K + 5, N, A), device=dev.device)
start = (ag.angles.clone(), ag.positions.clone())
imu_out = torch.zeros((N, A, 3), device=dev.device)
tbl = modules._table(mover._actionset)


def physics_variant(kind, i, state):
    if kind == 'plain':
        state['p'] = cuda.physics(sc, ag, out=state.get('p'))
    elif kind == 'move':
        state['p'] = cuda.physics(sc, ag, movement=(acts[i], tbl, 1 - mover.decay), out=state.get('p'))
    else:
        state['p'] = cuda.physics(sc, ag, movement=(acts[i], tbl, 1 - mover.decay), imu=(imu_out, imu.ang_scale, imu.speed_scale), out=state.get('p'))


def render_variant(kind, state):
    if kind == 'planes':
        state['r'] = cuda.render(sc, ag, out=state.get('r'))
    elif kind == 'obs':
        state['r'] = cuda.render(sc, ag, fields=(), pooled=dict(subsample=args.sub, max_depth=10., rgb=True, depth=True, centre=args.centre), out=state.get('r'))
    elif kind == 'obs+planes':
        state['r'] = cuda.render(sc, ag, pooled=dict(subsample=args.sub, max_depth=10., rgb=True, depth=True, centre=args.centre), out=state.get('r'))
    elif kind == 'obs-depth':
        state['r'] = cuda.render(sc, ag, fields=(), pooled=dict(subsample=args.sub, max_depth=10., rgb=False, depth=True), out=state.get('r'))
    elif kind == 'depth':
        state['r'] = cuda.render(sc, ag, fields=('distances',), out=state.get('r'))


def timed(pk, rk, reps=60):
    state = {}

    def rewind():
        ag.angles.copy_(start[0]); ag.positions.copy_(start[1]); ag.velocity.zero_(); ag.angvelocity.zero_()
        if pk == 'plain':
            ag.velocity.normal_(0, .5)
    rewind()
    for i in range(3):
        physics_variant(pk, i, state); render_variant(rk, state)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    rewind()
    with torch.cuda.graph(g):
        for i in range(K):
            physics_variant(pk, i, state); render_variant(rk, state)
    ts = []
    for _ in range(reps):
        rewind()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1)/K)
    return 1e3*float(np.median(ts))


for pk, rk in (('plain', 'planes'), ('move', 'planes'), ('move+imu', 'planes'), ('plain', 'obs'), ('plain', 'obs+planes'), ('plain', 'depth'),
               ('move+imu', 'obs'), ('plain', 'obs-depth')):
    print(f'physics {pk:9s} render {rk:11s}: {timed(pk, rk):7.2f} us per step', flush=True)
