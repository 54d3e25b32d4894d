"""Headline benchmark: env-steps/sec of the simulation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs 4096] [--agents 4] [--res 64]

Workload (BASELINE.json `metric`): 4096 envs x 4 agents x 64-ray RGBD per GPU, seeded synthetic cubicasa-like
floorplans, random momentum actions. One *step* = one pass of the hot path over the whole batch: `ms_physics` then
`ms_render`, both through the C-ABI; the step's inputs (velocities produced beforehand by the momentum-movement glue
from random actions) are resident in HBM and read in place. With --gpus N>1 the driver launches this file under torch.distributed.run; every rank owns its
own 4096-env slice (weak scaling, no data-path collective - envs are independent) and rank 0 prints one JSON line.

Besides the contract fields the line carries
  roofline      render kernel (the dominant one): algorithmic bytes per launch / its mean launch time (HIP events
                around every render launch of the timed region) against the 8 TB/s HBM peak;
  cpu_baseline  the CPU oracle (oracle/, a plain-C port of the reference's kernels) on this box's host cores, on a
                bounded sample of the same workload - reported only, never the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def build_world(n_envs, n_agents, res, fov, device, seed, n_unique=512, large=False):
    from megastep_amd import core, cubicasa, modules, scene
    np.random.seed(seed)
    pool = cubicasa.sample(min(n_unique, n_envs), seed=seed + 1, n_unique=max(n_unique, 16), large=large)
    geometries = [pool[i % len(pool)] for i in range(n_envs)]
    scenery = scene.scenery(geometries, n_agents, device=device, random=np.random.RandomState(seed))
    c = core.Core(scenery, res=res, fov=fov, fps=10)
    spawner = modules.RandomSpawns(geometries, c)
    torch.manual_seed(seed)
    spawner(c.agent_full(True))
    return c, geometries


def algorithmic_bytes(core):
    """Bytes one hot-path step must move, per SURVEY.md section 8(d) / BASELINE.md section 3, with the real ragged
    sizes of this scenery. Returns (render bytes per launch, physics bytes per launch)."""
    sc = core.scenery
    N, A, M, R = core.n_envs, core.n_agents, sc.model.shape[0], core.res
    L, I = sc.lines.vals.shape[0], sc.lights.vals.shape[0]
    render = (16*L + 12*I + 8*N            # lines, lights, ragged offsets read once per env
              + 12*N*A                     # angle + position read
              + 16*N*A*M                   # agent lines written back
              + N*A*R*(16 + 12)            # idx, loc, dot, dist + rgb written
              + N*A*R*40)                  # 2 texels x 12 B + 2 baked x 4 B + texture width/start gathered per ray
    physics = (16*(L - N*A*M) + 8*N        # wall segments + offsets read once per env
               + 48*N*A                    # agent state read + written
               + 4*N*A)                    # progress
    return render, physics


def env_step_fps(device, n_core_envs=4096, steps=60, warmup=10):
    """Whole env.step() rates - kernels plus the torch glue of megastep_amd.demo.envs - with random actions, the
    quantity the reference's docs quote (docs/index.rst:13-25: Explorer 180k FPS, Deathmatch 1.2m FPS on a 2080 Ti).
    Explorer renders 256 rays -> 64 px, Deathmatch 512 -> 128 px, as in the reference; FPS counts agent-envs."""
    from megastep_amd import arrdict, cubicasa
    from megastep_amd.demo import Deathmatch, Explorer
    pool = cubicasa.sample(256, n_unique=512)
    geometries = [pool[i % len(pool)] for i in range(n_core_envs)]

    def rate(env, n):
        env.reset()
        acts = torch.randint(0, 7, (steps + warmup, n, env.action_space.shape[0]), device=device)
        for i in range(warmup):
            env.step(arrdict.arrdict(actions=acts[i]))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            env.step(arrdict.arrdict(actions=acts[warmup + i]))
        torch.cuda.synchronize()
        eager = n*steps/(time.perf_counter() - t0)
        # the same step captured once in a HIP graph and replayed (possible because nothing in it syncs with the host)
        static = acts[0].clone()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            env.step(arrdict.arrdict(actions=static))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            static.copy_(acts[warmup + i])
            graph.replay()
        torch.cuda.synchronize()
        return eager, n*steps/(time.perf_counter() - t0)

    out = {}
    np.random.seed(0); torch.manual_seed(0)
    eager, graphed = rate(Explorer(n_core_envs, device=device, geometries=geometries), n_core_envs)
    out['explorer'] = {'fps': eager, 'fps_hip_graph': graphed,
                       'env': f'Explorer({n_core_envs}): 1 agent, 256 rays -> 64 px RGB+D+IMU'}
    torch.cuda.empty_cache()
    eager, graphed = rate(Deathmatch(4*n_core_envs, 4, device=device, geometries=geometries), 4*n_core_envs)
    out['deathmatch'] = {'fps': eager, 'fps_hip_graph': graphed,
                         'env': f'Deathmatch({4*n_core_envs}, 4): {n_core_envs} core envs x 4 agents, 512 rays -> 128 px RGB+D+IMU'}
    return out


def measured_traffic(args, world):
    """HBM bytes per ms_render launch from rocprofv3 PMC passes of this exact command (profiles/rNN_traffic.json,
    written by tools/profile.sh: FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950). None when the workload differs from the profiled one."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')), reverse=True):
        t = json.load(open(path))
        w = t.get('workload', {})
        if (w.get('envs'), w.get('agents'), w.get('res'), w.get('large', False)) == (args.envs, args.agents, args.res, args.large):
            return t['render_bytes_per_launch']
    return None


def cpu_baseline(core, budget_s=4., max_envs=4096):
    """The CPU oracle on a bounded sample: the first `max_envs` envs of this workload, all host cores (OpenMP over
    envs), for about `budget_s` seconds of wall time (a few tens of seconds of CPU work per 8 cores)."""
    from oracle import oracle as O
    from tests import util
    n = min(max_envs, core.n_envs)
    sc = core.scenery
    e_l, e_i = int(sc.lines.ends[n - 1]), int(sc.lights.ends[n - 1])
    e_t = int(sc.textures.ends[e_l - 1])
    g = lambda t: t.detach().cpu().numpy()
    scene = O.Scene(dict(
        n_agents=sc.n_agents, model=g(sc.model),
        lights_vals=g(sc.lights.vals[:e_i]), lights_widths=g(sc.lights.widths[:n]),
        lines_vals=g(sc.lines.vals[:e_l]), lines_widths=g(sc.lines.widths[:n]),
        textures_vals=g(sc.textures.vals[:e_t]), textures_widths=g(sc.textures.widths[:e_l]),
        baked_vals=g(sc.baked.vals[:e_t])))
    cfg = O.config(core.agent_radius, core.res, core.fov, core.fps)
    agents = {k: v[:n] for k, v in util.agents_dict(core.agents).items()}
    rng = np.random.RandomState(0)
    steps, t0 = 0, time.perf_counter()
    while True:
        agents['velocity'] = rng.uniform(-3, 3, agents['velocity'].shape).astype(np.float32)
        agents['angvelocity'] = rng.uniform(-180, 180, agents['angvelocity'].shape).astype(np.float32)
        _, agents = O.physics(scene, agents, cfg)
        O.render(scene, agents, cfg)
        steps += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or steps >= 100:
            break
    return {'value': n*steps/dt, 'unit': 'env-steps/s', 'cores': os.cpu_count(), 'kind': 'port',
            'sample': f'first {n} envs of the workload x {steps} steps (physics+render), C oracle with OpenMP over envs'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--envs', type=int, default=4096, help='envs per GPU')
    ap.add_argument('--agents', type=int, default=4)
    ap.add_argument('--res', type=int, default=64)
    ap.add_argument('--fov', type=float, default=130.)
    ap.add_argument('--large', action='store_true', help='800-1200 wall segments per env')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--env-fps', action='store_true',
                    help="also time full env.step() of the reference-shaped Explorer and Deathmatch envs (reported only)")
    args = ap.parse_args()

    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    distributed = world > 1
    assert torch.cuda.is_available(), 'bench.py needs a GPU: the product has no CPU path'
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=device)

    from megastep_amd import cuda, modules
    core, _ = build_world(args.envs, args.agents, args.res, args.fov, device, seed=1 + rank, large=args.large)
    N, A = core.n_envs, core.n_agents
    total = args.steps + args.warmup

    # pre-generated random momentum actions -> per-step velocity targets, resident in HBM
    torch.manual_seed(rank)
    mover = modules.MomentumMovement(core)
    actions = torch.randint(0, 7, (total, N, A), device=device)
    scenery, agents = core.scenery, core.agents

    def step(i):
        delta = mover._actionset[actions[i]]
        agents.angvelocity[:] = (1 - mover.decay)*agents.angvelocity + delta.angvelocity
        agents.velocity[:] = (1 - mover.decay)*agents.velocity + modules.to_global_frame(agents.angles, delta.velocity)
        cuda.physics(scenery, agents)
        return cuda.render(scenery, agents)

    # velocities per step are produced by the (untimed) torch movement glue ahead of time
    vel = torch.empty((total, N, A, 2), device=device)
    angvel = torch.empty((total, N, A), device=device)
    for i in range(total):       # a dry run of the env loop records the velocity targets
        step(i)
        vel[i], angvel[i] = agents.velocity, agents.angvelocity
    torch.cuda.synchronize()

    # One Agents view per step: positions/angles are the persistent state, velocity/angvelocity point at that
    # step's pre-generated inputs, already resident in HBM - the hot path reads its inputs in place, no copies.
    views = [cuda.Agents(agents.angles, agents.positions, angvel[i], vel[i]) for i in range(total)]

    def timed_step(i, ev=None):
        cuda.physics(scenery, views[i])
        if ev is not None:
            ev[0].record()
        r = cuda.render(scenery, views[i])
        if ev is not None:
            ev[1].record()
        return r

    for i in range(args.warmup):
        timed_step(i)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        timed_step(args.warmup + i, events[i])
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    from megastep_amd import sharding
    elapsed = sharding.max_over_ranks(elapsed, device)      # the slowest rank sets the step rate

    render_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
    rb, pb = algorithmic_bytes(core)
    achieved = rb/(render_ms*1e-3)/1e9
    ms_per_step = 1e3*elapsed/args.steps
    value = world*N*args.steps/elapsed

    out = {
        'metric': 'env-steps/sec', 'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {
            'workload': f'{N} envs x {A} agents x {args.res}-ray RGBD per GPU, fov {args.fov:g}, synthetic cubicasa floorplans'
                        + (' (large maps)' if args.large else ''),
            'step': 'ms_physics + ms_render (C-ABI), per-step velocities from random momentum actions resident in HBM',
            'envs_per_gpu': N, 'agents': A, 'res': args.res,
            'lines_per_env': scenery.lines.vals.shape[0]/N, 'lights_per_env': scenery.lights.vals.shape[0]/N,
            'parallelism': f'env-sharded x{world}, no collectives'},
        'agent_steps_per_sec': value*A,
        'roofline': {
            'kernel': 'ms_render = render_kernel<1,1,0> (headings cached by ms_physics)', 'bound': 'hbm', 'achieved': achieved,
            'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': achieved/HBM_PEAK_GBPS, 'traffic': measured_traffic(args, world),
            'algorithmic_bytes_per_launch': rb, 'avg_launch_ms': render_ms,
            'step_algorithmic_bytes': rb + pb, 'step_achieved_GBps': (rb + pb)/(ms_per_step*1e-3)/1e9},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(core)
    if rank == 0 and world == 1 and args.env_fps:
        del core, scenery, agents
        torch.cuda.empty_cache()
        out['env_step'] = env_step_fps(device)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
