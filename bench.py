"""Headline benchmark: env-steps/sec of the simulation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs 4096] [--agents 4] [--res 64]

Workload (BASELINE.json `metric`): 4096 envs x 4 agents x 64-ray RGBD per GPU, seeded synthetic cubicasa-like
floorplans, random momentum actions. One *step* = one pass of the hot path over the whole batch: `ms_physics` then
`ms_render`, both through the C-ABI; the step's inputs (velocities produced beforehand by the momentum-movement glue
from random actions) are resident in HBM and read in place.

The K timed steps are enqueued as ONE HIP graph (2K kernel nodes, recorded once, replayed inside the timed regions):
`value` is the replays' rate - what the kernels can do with the host out of the way. The same K steps launched one by
one from Python (`eager`) are timed beside it, with HIP events around every step and every render for the per-step
spread and the roofline.

With --gpus N>1 there is one rank per GPU: `python bench.py --gpus N` on its own starts the N ranks itself
(torch.distributed.run on this node, 127.0.0.1, a free port; it refuses if the node has fewer GPUs), and under a launcher
that has already set WORLD_SIZE (the driver's torch.distributed.run command) it is one of them. The job holds N x
--envs envs; every rank works out the same contiguous cuts (balanced by lines x agents x rays) from the floorplans on
the host and builds and bakes ITS slice only - weak scaling, no collective on the data path: envs are independent; the
ranks meet only in a gloo barrier BEFORE each timed region and a MAX over their own times after it (each rank's clock runs
from its own synchronize to its own synchronize: no rendezvous is inside what is timed), so RCCL is never initialised.
Rank 0 prints one JSON line.

Timing: a timed region is the W untimed warm-up steps from the spawn points and then exactly K steps between barrier +
synchronize pairs - steps W..W+K of one fixed trajectory of the env loop (recorded beforehand: the velocities handed to
`ms_physics` at every step), the same in every region. Regions are repeated until they add up to a quarter of a second
and `value` / `ms_per_step` are the MEDIAN region's (min and max beside it), so that a K = 20 run is as repeatable as a
K = 200 one and times the same kind of step.

Besides the contract fields the line carries
  roofline      render kernel (the dominant one): algorithmic bytes per launch / its mean launch time (HIP events
                around every render launch of the eager timed region) against the 8 TB/s HBM peak;
  cpu_baseline  a pure-PyTorch CPU step (oracle/torch_step.py, the restatement of the reference's kernels as tensor
                ops) on this box's host cores, on a bounded sample of the same workload - reported only, never the
                target; `cpu_baseline_c` is the plain-C port with OpenMP over envs on the same sample;
  shapes        BASELINE.json's other single-GPU shapes under the same protocol at K = 20, W = 5: C2 with all five planes
                and depth-only, C3, 512 rays, C5's per-GPU share - ms per step, env-steps/s, the render kernel's HIP-event
                median and its roofline fraction from that shape's own algorithmic bytes (N = 1 only);
  env_step      whole `env.step()` rates of the reference-shaped Explorer and Deathmatch envs (what the reference's
                docs quote), eager and replayed as a HIP graph; `env_step_headline_shape` the same at the headline shape.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# A knob of the HIP runtime, read when libamdhip64 loads (so: before torch is imported): kernel arguments of kernels
# launched one by one go into device memory instead of host memory that every wave's first scalar load then crosses the
# fabric for. Measured on boxes where the default is off: the eager leg's render launch 37.6 -> 34.9 us by HIP
# events, the eager step 58.7 -> 54.4 us, Deathmatch's eager env.step +15 %; the HIP-graph replays (`value`) keep their
# arguments in device memory either way and do not move; the host pays per launch for it (Explorer's eager env.step, which
# is bound by the host's launches, -18 %; its graph +3 %). Set it to 0 to see the other side.
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_VECTOR_PEAK_TFLOPS = 157.3   # same guide: 256 CUs x 2.4 GHz x 256 flop/clk (packed FMA), the VALU ceiling SURVEY 8(d) names
_T0 = time.perf_counter()
# processes that generate floorplans (forked before the GPU is touched). None under a profiler: rocprofv3's counter
# collection hangs on forked children (seen with --pmc), and a profile does not care how long the plans took
PLAN_WORKERS = 0 if any('rocprof' in os.environ.get(k, '').lower() for k in ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES', 'HSA_TOOLS_LIB')) \
    else min(os.cpu_count() or 1, 32)


# how those processes are started: 'fork' here (before the GPU is touched); tests that build benchmark worlds from inside a
# process that has long been using its GPU set 'subprocess' (fresh numpy-only interpreters: megastep_amd.cubicasa.prefetch)
PLAN_CONTEXT = 'fork'


def log(msg):
    """Progress on stderr (stdout carries the one JSON line)."""
    if int(os.environ.get('RANK', 0)) == 0:
        print(f'[bench {time.perf_counter() - _T0:7.1f}s] {msg}', file=sys.stderr, flush=True)


def plan_count(n_envs, n_agents, unique=None):
    """distinct floorplans = max(N // 4, 1), tiled, for a multi-agent world (as Deathmatch builds it,
    deathmatch.py:24), N for a single-agent one (Explorer, explorer.py:11) - unless --unique says otherwise."""
    if unique:
        return max(1, min(int(unique), n_envs))
    return n_envs if n_agents == 1 else max(n_envs//4, 1)


def world_geometries(n_envs, world, seed, n_unique=512, large=False, legacy=False, oblique=False):
    """The floorplan of every env of the whole (world x n_envs)-env job: a pool of exactly `n_unique` distinct plans, tiled.
    (`legacy`: the pool of rounds 1-3 - the training split of a 512-plan sample, 460 distinct - for a figure comparable
    with theirs; `oblique`: the plans turned by seeded angles, with diagonal partitions - megastep_amd/cubicasa.py.)  Plans that
    are not cached yet are generated on forked worker processes."""
    from megastep_amd import cubicasa
    workers = PLAN_WORKERS
    if legacy:
        pool = cubicasa.sample(min(512, n_envs), seed=seed + 1, n_unique=512, large=large, workers=workers, context=PLAN_CONTEXT)
    else:
        pool = cubicasa.sample(n_unique, split='all', seed=seed + 1, n_unique=n_unique, large=large, workers=workers, context=PLAN_CONTEXT,
                               oblique=oblique)
    return [pool[i % len(pool)] for i in range(world*n_envs)]


def rank_slice(geometries, n_agents, res, rank, world):
    """This rank's contiguous [start, stop) of the job's envs, balanced by lines x agents x rays - worked
    out on the host from the floorplans alone, before anything is built."""
    from megastep_amd import scene, sharding
    if world == 1:
        return 0, len(geometries)
    af = n_agents*len(scene.agent_model())
    cost = np.array([af + len(g['walls']) for g in geometries], np.float64)*n_agents*res
    return sharding.env_slice(len(geometries), rank, world, cost)


def build_world(n_envs, n_agents, res, fov, device, seed, n_unique=512, large=False, rank=0, world=1, bake=True, fast=False,
                legacy=False, oblique=False):
    """The benchmark's world - this rank's slice of it. With world > 1 the job has world x n_envs envs; every rank works
    out the same cost-balanced cuts from the floorplans and builds (and bakes) its own slice only: nothing of the other
    ranks' envs ever reaches this rank's device (reference: common.h:136-144 slices, it does not replicate)."""
    from megastep_amd import core, modules, scene
    geometries = world_geometries(n_envs, world, seed, n_unique, large, legacy, oblique)
    start, stop = rank_slice(geometries, n_agents, res, rank, world)
    np.random.seed(seed)
    scenery = scene.scenery(geometries, n_agents, device=device, random=np.random.RandomState(seed), bake=bake, fast=fast,
                            envs=(start, stop))
    geometries = geometries[start:stop]
    c = core.Core(scenery, res=res, fov=fov, fps=10)
    np.random.seed(seed + 1000*(rank + 1))
    spawner = modules.RandomSpawns(geometries, c, fast=fast)
    torch.manual_seed(seed + rank)
    spawner(c.agent_full(True))
    return c, geometries


def algorithmic_bytes(core, fields=None):
    """Bytes one hot-path step must move, per SURVEY.md section 8(d) / BASELINE.md section 3, with the real ragged
    sizes of this scenery. Returns (render bytes per launch, physics bytes per launch). `fields`: the per-ray outputs the
    render call is asked for (default: all five, RGBD); without `screen` neither the colour writes nor the texel /
    baked-light gathers are part of the job."""
    sc = core.scenery
    N, A, M, R = core.n_envs, core.n_agents, sc.model.shape[0], core.res
    L, I = sc.lines.vals.shape[0], sc.lights.vals.shape[0]
    fields = ('indices', 'locations', 'dots', 'distances', 'screen') if fields is None else fields
    per_ray = sum(12 if f == 'screen' else 4 for f in fields)     # idx, loc, dot, dist + rgb written
    if 'screen' in fields:
        per_ray += 40                                              # 2 texels x 12 B + 2 baked x 4 B + texture width/start gathered per ray
    render = (16*L + 12*I + 8*N            # lines, lights, ragged offsets read once per env
              + 12*N*A                     # angle + position read
              + 16*N*A*M                   # agent lines written back
              + N*A*R*per_ray)
    physics = (16*(L - N*A*M) + 8*N        # wall segments + offsets read once per env
               + 48*N*A                    # agent state read + written
               + 4*N*A)                    # progress
    rep = sc.grid_report().get('wall_grid') if hasattr(sc, 'grid_report') else None
    if rep and rep.get('cells'):
        # with a wall grid ms_physics does not stream the env's walls: an agent reads its cell's header and the rows of the
        # walls near the cell (the mean list over the grid's cells) - THAT is what the step asks of the memory system
        physics = N*A*(16 + 16*rep['near_rows']/rep['cells']) + 8*N + 52*N*A
    return render, physics


def all_pairs_flops(core):
    """What the reference's raycast costs per render launch, SURVEY.md section 8(d): every ray against every line of its
    env at ~30 flop a test, plus ~40 flop of shading per ray. The kernels here test a fraction of those pairs (a wave
    intersects ~7 lines per ray out of ~330), so dividing this by the kernel time gives an *all-pairs-equivalent* rate
    that may exceed the VALU peak; it is reported next to the HBM roofline as the survey asks, not as a utilisation."""
    sc = core.scenery
    A, R = core.n_agents, core.res
    return float(sc.lines.vals.shape[0])*A*R*30 + float(core.n_envs)*A*R*40


def env_step_fps(device, n_core_envs=4096, steps=40, warmup=8):
    """Whole env.step() rates - kernels plus the torch glue of megastep_amd.demo.envs - with random actions, the
    quantity the reference's docs quote (docs/index.rst:13-25: Explorer 180k FPS, Deathmatch 1.2m FPS on a 2080 Ti).
    Explorer renders 256 rays -> 64 px, Deathmatch 512 -> 128 px, as in the reference; FPS counts agent-envs."""
    from megastep_amd import arrdict
    from megastep_amd.demo import Deathmatch, Explorer
    # One distinct floorplan per core env, as the reference builds them: Explorer(n) samples n geometries (explorer.py:11),
    # Deathmatch(n, 4) samples n // 4 for its n // 4 core envs (deathmatch.py:24, where n counts agent-rows, :44) - out of its 4492.
    # (Rounds 2-5 tiled a 256-plan pool here: VERDICT r5, missing 2.)  The pool is C2's, generated before the GPU was touched.
    geometries = world_geometries(n_core_envs, 1, 1, n_core_envs)
    plans = len({id(g) for g in geometries})

    def rate(env, n):
        env.reset()
        acts = torch.randint(0, 7, (steps + warmup, n, env.action_space.shape[0]), device=device)
        for i in range(warmup):
            env.step(arrdict.arrdict(actions=acts[i]))
        torch.cuda.synchronize()
        times = []
        for _ in range(5):                                               # (the median of a few passes, as for the graphs)
            t0 = time.perf_counter()
            for i in range(steps):
                env.step(arrdict.arrdict(actions=acts[warmup + i]))
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        eager = n*steps/float(np.median(times))
        # the same steps as HIP graphs (possible because nothing in a step syncs with the host), actions taken by pointer
        return eager, n*steps/replay_steps(lambda a: env.step(arrdict.arrdict(actions=a)), acts[warmup:warmup + steps])

    out = {}
    np.random.seed(0); torch.manual_seed(0)
    from megastep_amd.demo import Minimal

    def leg(key, make, n, what, **more):
        """One env's rates; a leg that fails leaves its reason in the line and the others go on."""
        try:
            env = make()
            log(f'{key} built')
            eager, graphed = rate(env, n)
            del env
            out[key] = {'fps': eager, 'fps_hip_graph': graphed, 'env': what, **more}
        except Exception as e:                                             # noqa: BLE001
            log(f'env.step leg {key} FAILED: {type(e).__name__}: {e}')
            out[key] = {'fps': None, 'fps_hip_graph': None, 'env': what, 'error': f'{type(e).__name__}: {e}'[:400]}
        torch.cuda.empty_cache()

    leg('explorer', lambda: Explorer(n_core_envs, device=device, geometries=geometries), n_core_envs,
        f'Explorer({n_core_envs}): 1 agent, 256 rays -> 64 px RGB+D+IMU', distinct_floorplans=plans)
    # BASELINE config 2 says "depth-only": the same env without the RGB observation - the renderer's colourless instantiation
    leg('explorer_depth_only', lambda: Explorer(n_core_envs, device=device, geometries=geometries, depth_only=True), n_core_envs,
        f'Explorer({n_core_envs}, depth_only=True): 1 agent, 256 rays -> 64 px D+IMU', distinct_floorplans=plans)
    leg('minimal', lambda: Minimal(n_core_envs, device=device), n_core_envs,
        f"Minimal({n_core_envs}): the reference's tutorial env (one agent in a 5 m box, SimpleMovement, 64 rays RGB) - one launch a step "
        f"(ms_move_step_render)")
    leg('deathmatch', lambda: Deathmatch(4*n_core_envs, 4, device=device, geometries=geometries), 4*n_core_envs,
        f'Deathmatch({4*n_core_envs}, 4): {n_core_envs} core envs x 4 agents, 512 rays -> 128 px RGB+D+IMU', distinct_floorplans=plans)
    return out


def replay_steps(step, actions, repeats=5):
    """Seconds for len(actions) env steps replayed as HIP graphs: one graph per step, each captured on ITS OWN slice of the
    pre-drawn actions - the step reads its actions where they lie, as it would read a policy's output buffer. (Round 4 replayed
    one graph and copied every step's actions into its static input first: a 5 us copy kernel and a second launch per 45 us
    step, standing in for a policy's write that is not the env's to pay for.) The graphs share one memory pool, so they all
    write the same observation buffers; trajectories are the eager leg's, step for step."""
    graphs = []
    for a in actions:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=graphs[0].pool() if graphs else None):
            step(a)
        graphs.append(g)
    torch.cuda.synchronize()
    times = []
    for _ in range(repeats):                                             # (the median of a few passes: a pass is a few milliseconds)
        t0 = time.perf_counter()
        for g in graphs:
            g.replay()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    return float(np.median(times))


def traffic_entry(envs, agents, res, large=False, depth_only=False, plans=None, one_launch=False, oblique=False):
    """The newest profiles/rNN_traffic.json entry for a workload (tools/profile.sh: FETCH_SIZE and WRITE_SIZE in separate PMC
    passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; since round 5 also the vector ALUs' busy
    fraction from the SQ pass of the same profile) and the file it was read from; (None, None) for a shape nobody profiled -
    PMC counters cannot be collected from inside the benchmark process."""
    import glob
    want = (envs, agents, res, bool(large), bool(depth_only))
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')), reverse=True):
        t = json.load(open(path))
        for entry in t.get('shapes', [t]):                              # (one shape per file up to round 2, a list since)
            w = entry.get('workload', {})
            if (w.get('envs'), w.get('agents'), w.get('res'), bool(w.get('large', False)), bool(w.get('depth_only', False))) == want \
                    and bool(w.get('one_launch', False)) == bool(one_launch) and bool(w.get('oblique', False)) == bool(oblique):
                # (`plans`: the world's distinct floorplans - a profile of the same shape on another plan count moved other bytes:
                # the 64-plan C5 world lives in the caches, the 4096-plan one does not.  An entry that does not say is a profile of
                # the shape's default world, SURVEY 8(d)'s count; a line for another count gets no traffic rather than a wrong one.)
                have = w.get('plans', plan_count(envs, agents) if not large else C5_PLANS)
                if plans is None or have == plans:
                    return entry, os.path.relpath(path, ROOT)
    return None, None


def measured_traffic(args):
    """HBM bytes per ms_render launch of this exact workload from the rocprofv3 PMC passes (see traffic_entry)."""
    plans = 460 if args.legacy_plans else plan_count(args.envs, args.agents, args.unique)
    entry, path = traffic_entry(args.envs, args.agents, args.res, args.large, args.depth_only, plans, args.one_launch)
    return (entry['render_bytes_per_launch'], path) if entry else (None, None)


def measured_block(core, fields, render_ms, step_ms, large=False, one_launch=False, oblique=False):
    """What the counters say about a shape, next to its algorithmic-bytes roofline figure: the fabric traffic of the render
    launch and of the step, what fraction of the HBM peak THAT is over the measured time, and how busy the vector ALUs are -
    with the verdict on what bounds the shape. (The algorithmic formula of SURVEY 8(d) counts every line of an env once per
    launch; the kernels walk per-cell lists instead and at the larger shapes move LESS than it says - a fraction of bytes that
    were never moved is no utilisation, and round 4's C5 line read 8190 GB/s on an 8000 GB/s part that way.)"""
    sc = core.scenery
    plans = int((sc.geom == torch.arange(core.n_envs, device=sc.geom.device)).sum()) if getattr(sc, 'geom', None) is not None else core.n_envs
    entry, path = traffic_entry(core.n_envs, core.n_agents, core.res, large, fields is not None and 'screen' not in fields, plans, one_launch, oblique)
    if entry is None:
        return {'traffic': None, 'traffic_note': f'no profile of this shape on {plans} floorplans under profiles/'}
    rt, pt = entry['render_bytes_per_launch'], entry.get('physics_bytes_per_launch', 0.)
    out = {'traffic': rt, 'traffic_source': path + f" (shape '{entry.get('shape', '?')}', {plans} floorplans)",
           'frac_measured': rt/(render_ms*1e-3)/1e9/HBM_PEAK_GBPS,
           'step_traffic': rt + pt, 'step_measured_GBps': (rt + pt)/(step_ms*1e-3)/1e9}
    busy = entry.get('valu_busy_frac', {}).get('render_kernel')
    if busy is not None:
        out['valu_busy'] = busy
        out['bound_measured'] = ('vector ALU issue' if busy >= .85 else 'vector ALU issue, mostly' if busy >= .65 else
                                 "the launch's coarse grain: dependent round trips at both ends of a wave and the drain of the last waves")
    return out


def _oracle_sample(core, max_envs):
    """The first `max_envs` envs of the workload as an oracle scene dict + agents (numpy)."""
    from tests import util
    n = min(max_envs, core.n_envs)
    sc = core.scenery
    e_l, e_i = int(sc.lines.ends[n - 1]), int(sc.lights.ends[n - 1])
    e_t = int(sc.textures.ends[e_l - 1])
    g = lambda t: t.detach().cpu().numpy()
    scene = dict(
        n_agents=sc.n_agents, model=g(sc.model),
        lights_vals=g(sc.lights.vals[:e_i]), lights_widths=g(sc.lights.widths[:n]),
        lines_vals=g(sc.lines.vals[:e_l]), lines_widths=g(sc.lines.widths[:n]),
        textures_vals=g(sc.textures.vals[:e_t]), textures_widths=g(sc.textures.widths[:e_l]),
        baked_vals=g(sc.baked.vals[:e_t]))
    agents = {k: v[:n].copy() for k, v in util.agents_dict(core.agents).items()}
    return n, scene, agents


def cpu_baselines(core, budget_s=8.):
    """The reported-only CPU baselines on a bounded sample of the workload, on this box's host cores.

    `cpu_baseline`: the pure-PyTorch step (north_star; oracle/torch_step.py) on the first 256 envs, all cores through
    torch's intra-op threads. `cpu_baseline_c`: the plain-C oracle on the first 4096 envs, OpenMP over envs. Each runs
    for about `budget_s` seconds of wall time; both draw fresh random velocities every step, as the GPU legs' inputs do."""
    from oracle import oracle as O
    from oracle import torch_step
    cores = os.cpu_count()
    usable = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else cores
    out = {}
    rng = np.random.RandomState(0)

    # the pure-PyTorch step is thousands of small tensor ops: past a few dozen threads each op spends its time waking
    # threads up (256 of them turned one step into minutes), so the pool is capped; the sample is sized from a probe
    # step so that the leg stays within its budget whatever the box
    threads = max(1, min(usable, 32))
    torch.set_num_threads(threads)
    n = 16
    while True:
        n, scene, agents = _oracle_sample(core, n)
        world = torch_step.World(scene, core.agent_radius, core.res, core.fov, core.fps)
        ag = {k: torch.as_tensor(v) for k, v in agents.items()}
        t0 = time.perf_counter()
        torch_step.step(world, ag)
        probe = time.perf_counter() - t0
        log(f'pure-PyTorch CPU step: probe of {n} envs took {probe:.2f}s on {threads} threads')
        if n >= min(256, core.n_envs) or probe*4 > budget_s/4:
            break
        n *= 4
    steps, t0 = 0, time.perf_counter()
    while True:
        ag['velocity'] = torch.as_tensor(rng.uniform(-3, 3, agents['velocity'].shape).astype(np.float32))
        ag['angvelocity'] = torch.as_tensor(rng.uniform(-180, 180, agents['angvelocity'].shape).astype(np.float32))
        torch_step.step(world, ag)
        steps += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or steps >= 50:
            break
    log(f'pure-PyTorch CPU step: {steps} steps of {n} envs in {dt:.1f}s')
    out['cpu_baseline'] = {
        'value': n*steps/dt, 'unit': 'env-steps/s', 'cores': threads, 'kind': 'port', 'implementation': 'pure PyTorch (CPU tensors)',
        'host_cores': cores, 'usable_cores': usable, 'cores_used': threads,
        'why_not_all_cores': 'the step is thousands of small tensor ops: past a few dozen intra-op threads each op spends its time '
                             'waking threads up (256 threads turned one step into minutes); cpu_baseline_c uses every core',
        'sample': f'first {n} envs of the workload x {steps} steps (physics+render), oracle/torch_step.py, torch.set_num_threads({threads})'}

    # ... and the same step on EVERY host core (north_star: "the box's own host cores"), once, in a process of its
    # own with a deadline - in round 3 the 256-thread pool turned one step into minutes, which this run cannot afford to wait for
    if usable > threads:
        out['cpu_baseline']['all_cores'] = _torch_step_all_cores(core, usable, min(n, 16))

    n, scene, agents = _oracle_sample(core, 4096)
    scene = O.Scene(scene)
    cfg = O.config(core.agent_radius, core.res, core.fov, core.fps)
    steps, t0 = 0, time.perf_counter()
    while True:
        agents['velocity'] = rng.uniform(-3, 3, agents['velocity'].shape).astype(np.float32)
        agents['angvelocity'] = rng.uniform(-180, 180, agents['angvelocity'].shape).astype(np.float32)
        _, agents = O.physics(scene, agents, cfg)
        O.render(scene, agents, cfg)
        steps += 1
        dt = time.perf_counter() - t0
        if dt > budget_s/2 or steps >= 100:
            break
    out['cpu_baseline_c'] = {
        'value': n*steps/dt, 'unit': 'env-steps/s', 'cores': usable, 'kind': 'port', 'implementation': 'plain C + OpenMP over envs',
        'host_cores': cores,
        'sample': f'first {n} envs of the workload x {steps} steps (physics+render), oracle/megastep_oracle.c'}
    return out


_ALL_CORES = """
import pickle, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from oracle import torch_step
scene, agents, consts, threads = pickle.load(open(sys.argv[2], 'rb'))
torch.set_num_threads(threads)
world = torch_step.World(scene, *consts)
ag = {k: torch.as_tensor(v) for k, v in agents.items()}
torch_step.step(world, ag)                       # (thread pool and allocator warm)
rng = np.random.RandomState(0)
steps, t0 = 0, time.perf_counter()
while steps < 20 and time.perf_counter() - t0 < 6:
    ag['velocity'] = torch.as_tensor(rng.uniform(-3, 3, agents['velocity'].shape).astype(np.float32))
    ag['angvelocity'] = torch.as_tensor(rng.uniform(-180, 180, agents['angvelocity'].shape).astype(np.float32))
    torch_step.step(world, ag)
    steps += 1
print('RESULT', steps, time.perf_counter() - t0, flush=True)
"""


def _torch_step_all_cores(core, threads, n, deadline_s=25.):
    """The pure-PyTorch step with torch.set_num_threads(every usable core) on the first `n` envs, in a subprocess that is given
    `deadline_s` seconds: its rate, or the statement that it did not get there."""
    import pickle
    import subprocess
    import tempfile
    n, scene, agents = _oracle_sample(core, n)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'sample.pkl')
        pickle.dump((scene, agents, (core.agent_radius, core.res, core.fov, core.fps), threads), open(path, 'wb'), protocol=4)
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
        try:
            got = subprocess.run([sys.executable, '-c', _ALL_CORES, ROOT, path], capture_output=True, text=True, timeout=deadline_s, env=env)
            line = [l for l in got.stdout.splitlines() if l.startswith('RESULT')]
            if line:
                steps, dt = int(line[0].split()[1]), float(line[0].split()[2])
                log(f'pure-PyTorch CPU step on {threads} threads: {steps} steps of {n} envs in {dt:.1f}s')
                return {'value': n*steps/dt, 'unit': 'env-steps/s', 'cores': threads,
                        'sample': f'first {n} envs x {steps} steps, torch.set_num_threads({threads}), own process'}
            return {'value': None, 'cores': threads, 'note': 'the all-cores run failed: ' + got.stderr[-200:]}
        except Exception as e:                                             # (a reported-only leg never takes the line down with it)
            if not isinstance(e, subprocess.TimeoutExpired):
                return {'value': None, 'cores': threads, 'note': f'the all-cores run could not be made: {type(e).__name__}: {e}'[:300]}
            log(f'pure-PyTorch CPU step on {threads} threads: not through its first steps of {n} envs after {deadline_s:.0f}s')
            return {'value': None, 'cores': threads, 'upper_bound': n*2/deadline_s,
                    'note': f'with torch.set_num_threads({threads}) the step of {n} envs did not get through two steps (one to warm up) in '
                            f'{deadline_s:.0f} s: thousands of small tensor ops, each waking {threads} threads'}


class _Gpu:
    """What the timing harness needs from the device; `_Stub` below stands in for it in the CPU dry run."""

    def __init__(self, local_rank):
        assert torch.cuda.is_available(), 'bench.py needs a GPU: the product has no CPU path'
        torch.cuda.set_device(local_rank)
        self.device = torch.device('cuda', local_rank)

    def sync(self):
        torch.cuda.synchronize()

    def event(self):
        return torch.cuda.Event(enable_timing=True)

    def hot_path(self, scenery, fields=None, one_launch=False):
        from megastep_amd import cuda
        state = {}

        def fused(view, ev=None):
            # ms_step_render: the step as ONE launch (single-agent worlds of up to 64 rays) - the "render" events then
            # bracket the whole step
            if ev is not None:
                ev[1].record()
            state['pr'] = cuda.step_render(scenery, view, fields=fields, out=state.get('pr'))
            if ev is not None:
                ev[2].record()
        if one_launch:
            return fused

        def step(view, ev=None):
            state['p'] = cuda.physics(scenery, view, out=state.get('p'))
            if ev is not None:
                ev[1].record()
            state['r'] = cuda.render(scenery, view, fields=fields, out=state.get('r'))
            if ev is not None:
                ev[2].record()
        return step

    def graph(self, fn):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return g.replay


class _Stub:
    """CPU stand-in for the device (python bench.py --dry-run-cpu, used by tests/test_bench_gloo.py): the world is built
    and sharded for real on CPU tensors, the two kernel calls are replaced by a token tensor op, events by wall-clock
    stamps. Exercises everything around the kernels - arguments, rendezvous, slices, timing, the JSON line."""

    device = torch.device('cpu')

    def sync(self):
        pass

    def event(self):
        class E:
            def record(self):
                self.t = time.perf_counter()

            def elapsed_time(self, other):
                return 1e3*(other.t - self.t)
        return E()

    def hot_path(self, scenery, fields=None, one_launch=False):
        def step(view, ev=None):
            view.positions.add_(view.velocity, alpha=.1)
            if ev is not None:
                ev[1].record()
            view.angles.add_(view.angvelocity, alpha=.1)
            if ev is not None:
                ev[2].record()
        return step

    def graph(self, fn):
        return fn


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside any launcher: starts the N ranks itself - one process per GPU (the reference's
    model: one device per process, common.h:39-41) under torch.distributed.run on this node, rendezvous on 127.0.0.1 at a
    free port, rank r on device r - and passes rank 0's JSON line through. Refuses loudly when the node has fewer GPUs."""
    import socket
    import subprocess
    if not args.dry_run_cpu:
        have = torch.cuda.device_count()
        if have < args.gpus and not (args.share_gpu and have >= 1):
            raise SystemExit(f'bench.py: --gpus {args.gpus} asked for, but this node shows {have} GPU(s)')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS=os.environ.get('OMP_NUM_THREADS', '8'))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    log(f'starting {args.gpus} ranks: {" ".join(cmd[1:8])} ...')
    raise SystemExit(subprocess.call(cmd, env=env, cwd=ROOT))


def time_hot_path(dev, core, steps, warmup, barrier=lambda: None, rank=0, fields=None, graph=True, eager_floor=None, graph_floor=.25,
                  one_launch=False):
    """The timing protocol (module docstring) on one world: a dry run of the env loop records the W + K steps' inputs;
    the K timed steps are then launched one by one with HIP events around every step and every render (`eager`) and, with
    `graph`, replayed as one HIP graph - timed regions repeated until they add up to the floors. Returns the raw numbers."""
    from megastep_amd import cuda, modules, sharding
    device = dev.device
    N, A = core.n_envs, core.n_agents
    total = steps + warmup
    # pre-generated random momentum actions -> per-step velocity targets, resident in HBM
    torch.manual_seed(rank)
    mover = modules.MomentumMovement(core)
    actions = torch.randint(0, 7, (total, N, A), device=device)
    scenery, agents = core.scenery, core.agents
    hot = dev.hot_path(scenery, fields, one_launch)

    # The velocities the hot path is handed at each step are produced by the (untimed) torch movement glue ahead of
    # time: a dry run of the env loop from the spawn points, W + K steps, recording what ms_physics is given - the
    # velocities BEFORE it stops the agents that run into something.
    start = (agents.angles.clone(), agents.positions.clone())
    vel0 = torch.empty((total, N, A, 2), device=device)
    angvel0 = torch.empty((total, N, A), device=device)
    for i in range(total):
        delta = mover._actionset[actions[i]]
        agents.angvelocity[:] = (1 - mover.decay)*agents.angvelocity + delta.angvelocity
        agents.velocity[:] = (1 - mover.decay)*agents.velocity + modules.to_global_frame(agents.angles, delta.velocity)
        vel0[i], angvel0[i] = agents.velocity, agents.angvelocity
        hot(agents)
    dev.sync()

    # One Agents view per step: positions/angles are the persistent state, velocity/angvelocity point at that
    # step's pre-generated inputs, already resident in HBM - the hot path reads its inputs in place, no copies.
    vel, angvel = vel0.clone(), angvel0.clone()                         # (ms_physics zeroes the stopped agents' in place)
    views = [cuda.Agents(agents.angles, agents.positions, angvel[i], vel[i], config=core.config) for i in range(total)]

    def rewind():
        """Back to the spawn points, inputs pristine, then the W untimed warm-up steps: every timed region is the same K
        steps of the same trajectory - steps W..W+K of the env loop dry-run above - however often it is repeated.  (Regions
        that carried on from each other's end state replayed K inputs thousands of times over, agents ground into the
        corners their net drift pointed at: with the driver's --steps 20 a third slower than with --steps 200.)"""
        agents.angles.copy_(start[0]); agents.positions.copy_(start[1])
        vel.copy_(vel0); angvel.copy_(angvel0)
        for i in range(warmup):
            hot(views[i])

    own = []

    def timed(run):
        """One timed region: W untimed warm-up steps from the spawn points, then exactly K steps between barrier +
        synchronize pairs; the slowest rank's wall time."""
        rewind()
        barrier()
        dev.sync()
        t0 = time.perf_counter()
        run()
        dev.sync()
        own.append(time.perf_counter() - t0)                           # this rank's own sync-to-sync time: the clock stops HERE
        # The slowest rank's OWN K steps set the job's rate (N / max_g t_g after a start barrier). Nothing that
        # the ranks do to meet again - the MAX below is itself a gloo round trip, a closing barrier would be another - is
        # inside what is timed: at the driver's --steps 20 a region lasts under a millisecond, and round 4's closing
        # barrier (half a millisecond of TCP on 2 ranks) would have read as a 5x ceiling on the 8-GPU line.
        return sharding.max_over_ranks(own[-1])

    def repeated(run, floor_s=0.25, least=5, most=400):
        """`run` (exactly K steps) timed again and again - every region bracketed as above - until the regions add up to
        `floor_s` seconds: with the driver's --steps 20 one region lasts a millisecond, too short to be a measurement on
        its own. All ranks repeat equally often (the count comes from the first region's MAX over ranks)."""
        first = timed(run)
        n = int(min(max(np.ceil(floor_s/max(first, 1e-9)), least), most))
        return np.array([first] + [timed(run) for _ in range(n - 1)])

    # ---- eager: K steps launched one by one, events around every step and every render
    events = []

    def eager():
        evs = [(dev.event(), dev.event(), dev.event()) for _ in range(steps)]
        for i in range(steps):
            evs[i][0].record()
            hot(views[warmup + i], evs[i])
        events.append(evs)
    if eager_floor is None:
        eager_floor = 0.1 if steps < 100 else 0.
    eager_runs = repeated(eager, floor_s=eager_floor)
    m = {'eager_runs': eager_runs,
         'step_ms': np.array([e[0].elapsed_time(e[2]) for evs in events for e in evs]),
         'render_each': np.array([e[1].elapsed_time(e[2]) for evs in events for e in evs]),
         'graph_runs': None}
    # ---- graph: the same K steps as one HIP graph, replayed
    if graph:
        replay = dev.graph(lambda: [hot(views[warmup + i]) for i in range(steps)])
        replay()                                                       # (instantiation / first-launch costs stay outside)
        m['graph_runs'] = repeated(replay, floor_s=graph_floor)
    runs = m['graph_runs'] if m['graph_runs'] is not None else eager_runs
    m['runs'] = runs
    m['own_ms_per_step'] = 1e3*float(np.median(own[-len(runs):]))/steps
    return m


def ray_groups(dev, core):
    """render_kernel's NG for this world: what this thread's last ms_render actually launched (ms_debug_last_render_groups - the
    library's own record, with its pins, the scenery's grids and the device's wave slots taken into account; ADVICE r5: round 5
    re-derived the rule here and could label a line with a kernel that was not the one launched). Every caller has just timed
    the world's hot path on this thread; the CPU dry run, which launches nothing, asks the library's plan."""
    import ctypes
    from megastep_amd import _lib
    if dev.device.type == 'cuda':
        return int(_lib.lib().ms_debug_last_render_groups())
    ng = ctypes.c_int(0)
    _lib.lib().ms_host_render_plan(core.n_envs, core.n_agents, core.res, 4*6*256, 0, -1., -1, ctypes.byref(ng))
    return ng.value


def grids(core):
    """What bake() built around the floorplans (Scenery.grid_report()): bytes, cell sizes actually used, floorplans - a wall grid
    that outgrew its budget coarsens itself, which costs the step 3-6 %: the line says so instead of leaving it to be guessed."""
    rep = core.scenery.grid_report() if hasattr(core.scenery, 'grid_report') else {}
    out = {}
    for k in ('wall_grid', 'light_grid'):
        r = rep.get(k)
        if r:
            out[k] = {kk: r[kk] for kk in ('bytes', 'bytes_per_floorplan', 'cell', 'cells', 'floorplans', 'coarsened', 'candidate_rows', 'budget', 'vis_entries', 'near_rows') if kk in r}
    if rep.get('bake_seconds'):
        out['bake_seconds'] = rep['bake_seconds']                        # (one-off: ms_bake with its light grid; the wall grid's levels)
    return out


def shape_entry(dev, core, steps, warmup, fields=None, note=None, one_launch=False):
    """One line of the `shapes` block: the hot path on another of BASELINE.json's shapes, timed like the headline (same
    protocol, shorter floors), with that shape's own algorithmic bytes against the HBM peak."""
    m = time_hot_path(dev, core, steps, warmup, fields=fields, eager_floor=.05, graph_floor=.12, one_launch=one_launch)
    s = float(np.median(m['runs']))
    render_ms = float(np.median(m['render_each']))
    rb, pb = algorithmic_bytes(core, fields)
    sc = core.scenery
    e = {'envs': core.n_envs, 'agents': core.n_agents, 'res': core.res, 'fov': core.fov,
         'outputs': 'RGBD (all five planes)' if fields is None else '+'.join(fields),
         'lines_per_env': sc.lines.vals.shape[0]/core.n_envs,
         'distinct_floorplans': int((sc.geom == torch.arange(core.n_envs, device=sc.geom.device)).sum()) if sc.geom is not None else core.n_envs,
         'ms_per_step': 1e3*s/steps, 'env_steps_per_s': core.n_envs*steps/s, 'timed_regions': int(len(m['runs'])),
         'eager_ms_per_step': 1e3*float(np.median(m['eager_runs']))/steps,
         'render_launch_ms': render_ms, 'render_algorithmic_bytes': rb,
         'roofline_frac': rb/(render_ms*1e-3)/1e9/HBM_PEAK_GBPS, 'physics_algorithmic_bytes': pb,
         'roofline_frac_note': 'render_algorithmic_bytes (SURVEY 8(d): every line of an env once per launch, whether the kernel '
                               'reads it or not) over the render launch against 8 TB/s; what was actually moved: traffic / frac_measured',
         **measured_block(core, fields, render_ms, 1e3*s/steps, large=sc.lines.vals.shape[0]/core.n_envs > 600, one_launch=one_launch,
                          oblique=getattr(core, 'oblique', False)),
         'ray_groups_per_wave': ray_groups(dev, core), **grids(core),
         **({'world_build_seconds': core.build_seconds} if hasattr(core, 'build_seconds') else {})}
    if one_launch:
        from megastep_amd import _lib
        e['launches_per_step'] = 1 if _lib.lib().ms_debug_last_step_fused() else 2
        e['step'] = 'ms_step_render (physics + render of an env as one wave: one launch a step); render_launch_ms is the whole step'
        # (the one launch does both halves' work: its algorithmic bytes are the step's)
        e['roofline_frac'] = (rb + pb)/(render_ms*1e-3)/1e9/HBM_PEAK_GBPS
    if note:
        e['note'] = note
    return e


C5_PLANS = 4096          # distinct large floorplans of C5's per-GPU share (the reference's pool: 4492)


def other_shapes(dev, steps=20, warmup=5):
    """BASELINE.json's other single-GPU shapes, each timed with the headline's protocol at the driver's K / W: C2 (RGBD and
    depth-only), C3, the reference Deathmatch's own 512 rays, C5's per-GPU share, and the headline on rounds 1-3's
    460-plan pool for continuity. About 20 s altogether."""
    from megastep_amd import core as core_
    out = {}

    def world(tag, *a, **kw):
        t0 = time.perf_counter()
        c, _ = build_world(*a, device=dev.device, seed=1, **kw)
        c.build_seconds = time.perf_counter() - t0                       # (floorplans cached; scene assembly + bake + grids + spawns)
        log(f'{tag}: world built in {c.build_seconds:.1f}s')
        return c

    def guard(tag, fn):
        """One world's shapes: if it fails (a GPU too small or too busy for C5's grid, say) the line says so under
        `<tag>_error` and carries on with the next world."""
        try:
            fn()
        except Exception as e:                                             # noqa: BLE001
            log(f'shapes: {tag} FAILED: {type(e).__name__}: {e}')
            out[tag + '_error'] = f'{type(e).__name__}: {e}'[:400]
        torch.cuda.empty_cache()

    def _c2():
        c = world('C2', 4096, 1, 64, 130., n_unique=plan_count(4096, 1))
        out['c2_rgbd'] = shape_entry(dev, c, steps, warmup, note='BASELINE config 2 (Explorer shape, one floorplan per env) with all five planes')
        out['c2_depth_only'] = shape_entry(dev, c, steps, warmup, fields=('distances',),
                                           note="BASELINE config 2 as stated: depth-only - render_kernel<2,1,1,0,1>, no shading pass")
        out['c2_rgbd_one_launch'] = shape_entry(dev, c, steps, warmup, one_launch=True,
                                                note='BASELINE config 2 with all five planes, the step as ONE launch (ms_step_render)')
        out['c2_depth_only_one_launch'] = shape_entry(dev, c, steps, warmup, fields=('distances',), one_launch=True,
                                                      note='BASELINE config 2 as stated (depth-only), the step as ONE launch: render_kernel<2,1,1,0,1,1>')
    guard('c2', _c2)

    def _c3_r512():
        c = world('C3', 4096, 4, 128, 70., n_unique=plan_count(4096, 4))
        out['c3'] = shape_entry(dev, c, steps, warmup, note='BASELINE config 3 (Deathmatch shape at 128 rays)')
        c512 = core_.Core(c.scenery, res=512, fov=70., fps=10)
        c512.agents.positions[:], c512.agents.angles[:] = c.agents.positions, c.agents.angles
        out['r512'] = shape_entry(dev, c512, steps, warmup, note="the reference Deathmatch's own resolution (512 rays -> 128 px)")
    guard('c3_r512', _c3_r512)

    def _c5():
        c = world('C5 share', 32768, 1, 256, 130., n_unique=C5_PLANS, large=True, fast=True)
        out['c5_per_gpu_share'] = shape_entry(dev, c, steps, warmup, note=f'BASELINE config 5 / 8 GPUs: 32768 envs of 800-1200 walls on {C5_PLANS} distinct '
                                              'plans tiled - the diversity of the reference, whose cubicasa.sample tiles 4492 geometries '
                                              '(megastep/cubicasa.py:177-224); wall grid un-coarsened (see wall_grid)')
    guard('c5', _c5)

    def _c5_64():
        c = world('C5 share, 64 plans', 32768, 1, 256, 130., n_unique=64, large=True, fast=True)
        out['c5_per_gpu_share_64_plans'] = shape_entry(dev, c, steps, warmup, note="the same on rounds 3-4's 64 distinct plans (a 0.5 GB wall grid that "
                                                       "lives in the caches): for continuity, not the figure of record")
    guard('c5_64', _c5_64)

    def _headline_4096():
        c = world('headline, 4096 plans', 4096, 4, 64, 130., n_unique=4096)
        out['headline_4096_plans'] = shape_entry(dev, c, steps, warmup, note="the headline shape on ONE DISTINCT FLOORPLAN PER CORE ENV - what the reference's "
                                                 "Deathmatch(16384, 4) builds (deathmatch.py:24: cubicasa.sample(n_envs // 4) for its n_envs // 4 core envs; "
                                                 "SURVEY 8(d)'s 'N // 4 tiled', which the headline follows, reads that line as a quarter of that)")
    guard('headline_4096', _headline_4096)

    def _headline_oblique():
        c = world('headline, oblique plans', 4096, 4, 64, 130., n_unique=plan_count(4096, 4), oblique=True)
        c.oblique = True
        out['headline_oblique'] = shape_entry(dev, c, steps, warmup, note="the headline shape on floorplans turned by seeded angles, with diagonal partitions "
                                              "(cubicasa.sample(oblique=True)): the reference's walls are exteriors of arbitrary SVG polygons "
                                              "(geometry.py:43-57), the synthetic generator's are axis-aligned - same plan count as the headline")
    guard('headline_oblique', _headline_oblique)

    def _headline_460():
        c = world('headline, 460 plans', 4096, 4, 64, 130., legacy=True)
        out['headline_460_plans'] = shape_entry(dev, c, steps, warmup, note="the headline shape on rounds 1-3's floorplan pool (the training "
                                                "split of a 512-plan sample), for continuity with BENCH_r01..r03")
    guard('headline_460', _headline_460)

    return out


def headline_env_step(dev, core, steps=40, warmup=8):
    """Whole env.step() at the HEADLINE shape: random actions -> MomentumMovement + ms_physics (one launch, IMU reading
    included) -> ms_render with the RGB, depth observations pooled by the kernel (subsample 1) -> obs dict; eager and as a
    HIP graph. What an RL loop on this shape pays per step, next to the kernels-only `value`."""
    from megastep_amd import arrdict, modules
    mover, rgb, depth, imu = modules.MomentumMovement(core), modules.RGB(core), modules.Depth(core), modules.IMU(core)
    N, A = core.n_envs, core.n_agents

    def step(actions):
        mover(arrdict.arrdict(actions=actions), imu=imu)
        frame = modules.render(core, observers=(rgb, depth), fields=())
        return arrdict.arrdict(rgb=rgb(frame), d=depth(frame), imu=imu())
    acts = torch.randint(0, 7, (steps + warmup, N, A), device=dev.device)
    for i in range(warmup):
        step(acts[i])
    dev.sync()
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(steps):
            step(acts[warmup + i])
        dev.sync()
        times.append(time.perf_counter() - t0)
    eager = float(np.median(times))/steps
    graphed = replay_steps(step, acts[warmup:warmup + steps])/steps
    return {'what': f'{N} envs x {A} agents x {core.res} rays: MomentumMovement + physics (+ IMU) in one launch, render with RGB + depth '
                    'observations written by the kernel, random actions',
            'ms_per_step': 1e3*eager, 'env_steps_per_s': N/eager, 'ms_per_step_hip_graph': 1e3*graphed, 'env_steps_per_s_hip_graph': N/graphed,
            'hip_graph': 'one graph per step, each reading its own slice of the pre-drawn actions in place (no copy into a static input)'}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--envs', type=int, default=4096, help='envs per GPU')
    ap.add_argument('--agents', type=int, default=4)
    ap.add_argument('--res', type=int, default=64)
    ap.add_argument('--fov', type=float, default=130.)
    ap.add_argument('--large', action='store_true', help='800-1200 wall segments per env')
    ap.add_argument('--unique', type=int, default=0, help='distinct floorplans per GPU, tiled (default, SURVEY 8(d): envs/4 for a '
                                                         'multi-agent world, one per env for a single-agent one)')
    ap.add_argument('--legacy-plans', action='store_true', help="rounds 1-3's pool: the training split of a 512-plan sample (460 plans)")
    ap.add_argument('--depth-only', action='store_true', help="ask the renderer for `distances` alone (BASELINE config 2)")
    ap.add_argument('--one-launch', action='store_true', help="the step through ms_step_render: one launch a step for single-agent worlds of up "
                                                            "to 64 rays (BASELINE config 2), the two launches for every other shape")
    ap.add_argument('--fast-build', action='store_true', help="draw textures, lights and spawns on the device (10^4+ envs)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-env-fps', action='store_true', help="skip the whole-env.step() rates")
    ap.add_argument('--no-shapes', action='store_true', help="skip the other BASELINE shapes (the `shapes` block)")
    ap.add_argument('--env-fps', action='store_true', help='(default now; kept for old command lines)')
    ap.add_argument('--no-graph', action='store_true', help='value = the eager leg (no HIP graph)')
    ap.add_argument('--dry-run-cpu', action='store_true', help='no GPU: stub kernels, real plumbing (tests)')
    ap.add_argument('--plan-workers', type=int, default=None, help='processes generating floorplans (0: in this process)')
    ap.add_argument('--share-gpu', action='store_true', help='plumbing check on a box with fewer GPUs than ranks: rank r runs on device '
                                                          'r mod the devices there are (the rates mean nothing then, and the line says so)')
    ap.add_argument('--plan-cache', default=None, help='pickle of generated floorplans: loaded if it exists, written (and the run ended) with '
                                                       '--plans-only - for profiled runs, which cannot fork plan workers')
    ap.add_argument('--plans-only', action='store_true')
    ap.add_argument('--baseline-line', default=None, help="file holding the JSON line of the same command at --gpus 1: the line then "
                                                          "carries scaling_efficiency = value / (N x that line's value)")
    args = ap.parse_args(argv)
    if args.plan_workers is not None:
        global PLAN_WORKERS
        PLAN_WORKERS = args.plan_workers
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(args, sys.argv[1:] if argv is None else list(argv))

    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    if world != max(args.gpus, 1):
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU '
                         f'(python bench.py --gpus N starts them itself)')
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    distributed = world > 1
    import faulthandler
    faulthandler.enable()
    if os.environ.get('BENCH_WATCHDOG_S'):       # where is it, if it is still running by then?
        faulthandler.dump_traceback_later(float(os.environ['BENCH_WATCHDOG_S']), repeat=True)
    n_unique = plan_count(args.envs, args.agents, args.unique)
    extras = rank == 0 and world == 1 and not args.dry_run_cpu
    # Floorplans first, on forked workers, before this process touches its GPU: the headline's pool (which the C3 / 512-ray
    # shapes share) and, if the `shapes` block is wanted, C2's one-plan-per-env pool and the large maps.
    if args.plan_cache and os.path.exists(args.plan_cache):
        from megastep_amd import cubicasa
        cubicasa.load_cache(args.plan_cache)
    world_geometries(args.envs, 1, 1, n_unique, args.large, args.legacy_plans)
    if args.plans_only:
        from megastep_amd import cubicasa
        cubicasa.save_cache(args.plan_cache)
        return None
    if extras and not args.no_shapes:
        world_geometries(4096, 1, 1, 4096)
        world_geometries(4096, 1, 1, C5_PLANS, large=True)
        world_geometries(4096, 1, 1, 64, large=True)
        world_geometries(4096, 1, 1, legacy=True)
        world_geometries(4096, 1, 1, plan_count(4096, 4), oblique=True)
    elif extras and not args.no_env_fps:
        world_geometries(4096, 1, 1, 4096)                               # (the env.step legs': one plan per core env)
    log('floorplans ready')
    dev = _Stub() if args.dry_run_cpu else _Gpu(local_rank % max(torch.cuda.device_count(), 1) if args.share_gpu else local_rank)
    device = dev.device
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # gloo: the ranks only meet in a barrier and a MAX of one float - envs are independent, RCCL stays out of it
        dist.init_process_group('gloo')
        barrier = dist.barrier
        if os.environ.get('BENCH_TEST_BARRIER_SLEEP_MS'):
            # test hook (tests/test_bench_gloo.py): a slow rendezvous must not show in `value` - only what a rank does
            # between its own two synchronize calls is timed
            def barrier(_ms=float(os.environ['BENCH_TEST_BARRIER_SLEEP_MS'])):
                time.sleep(_ms*1e-3)
                dist.barrier()
    else:
        barrier = lambda: None

    core, _ = build_world(args.envs, args.agents, args.res, args.fov, device, seed=1, n_unique=n_unique, large=args.large,
                          rank=rank, world=world, bake=not args.dry_run_cpu, fast=args.fast_build, legacy=args.legacy_plans)
    N, A = core.n_envs, core.n_agents
    scenery = core.scenery
    log(f'world built: {N} envs on this rank')
    fields = ('distances',) if args.depth_only else None
    m = time_hot_path(dev, core, args.steps, args.warmup, barrier, rank, fields=fields, graph=not args.no_graph, one_launch=args.one_launch)
    eager_runs, graph_runs, runs = m['eager_runs'], m['graph_runs'], m['runs']
    step_ms, render_each = m['step_ms'], m['render_each']
    eager_s = float(np.median(eager_runs))
    render_ms = float(np.median(render_each))
    log(f'eager leg: {1e3*eager_s/args.steps:.4f} ms/step (median of {len(eager_runs)} regions of {args.steps} steps)')
    graph_s = None
    if graph_runs is not None:
        graph_s = float(np.median(graph_runs))
        log(f'graph leg: {1e3*graph_s/args.steps:.4f} ms/step (median of {len(graph_runs)} replays of {args.steps} steps)')

    elapsed = graph_s if graph_s is not None else eager_s
    # every rank's own median region of the leg `value` comes from (before it waits for the others): an imbalance shows here
    per_rank = [(N, m['own_ms_per_step'])]
    n_total = N
    if distributed:
        gathered = [None]*world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
        n_total = sum(n for n, _ in per_rank)
    rb, pb = algorithmic_bytes(core, fields)
    flops = all_pairs_flops(core)
    achieved = rb/(render_ms*1e-3)/1e9
    ms_per_step = 1e3*elapsed/args.steps
    value = n_total*args.steps/elapsed
    traffic, traffic_source = measured_traffic(args)
    outputs = 'depth-only (distances)' if args.depth_only else 'RGBD'

    out = {
        'metric': 'env-steps/sec', 'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {
            'workload': f'{args.envs} envs x {A} agents x {args.res}-ray {outputs} per GPU, fov {args.fov:g}, synthetic cubicasa floorplans'
                        + (' (large maps)' if args.large else ''),
            'step': ('ms_step_render (C-ABI: physics + render as one launch where an agent is one wave, else the two launches)' if args.one_launch
                     else 'ms_physics + ms_render (C-ABI)') + ', per-step velocities from random momentum actions resident in HBM',
            'launch': 'the K timed steps replayed as one HIP graph' if graph_s is not None else 'one Python call per kernel (eager)',
            'hip_force_dev_kernarg': os.environ.get('HIP_FORCE_DEV_KERNARG'),
            'envs_per_gpu': args.envs, 'envs_this_rank': N, 'envs_total': n_total, 'agents': A, 'res': args.res,
            'lines_per_env': scenery.lines.vals.shape[0]/N, 'lights_per_env': scenery.lights.vals.shape[0]/N,
            'distinct_floorplans_per_gpu': 460 if args.legacy_plans else n_unique, **grids(core),
            'parallelism': f'env-sharded x{world} (contiguous slices balanced by lines x agents x rays), no collectives'},
        'agent_steps_per_sec': value*A,
        'per_rank': {'envs': [n for n, _ in per_rank], 'ms_per_step': [t for _, t in per_rank]},
        # `value` is the MEDIAN timed region (each exactly K steps between barrier + synchronize pairs); the spread:
        'timed_regions': {'count': int(len(runs)), 'ms_per_step_min': 1e3*float(runs.min())/args.steps,
                          'ms_per_step_median': ms_per_step, 'ms_per_step_max': 1e3*float(runs.max())/args.steps},
        'eager': {'value': n_total*args.steps/eager_s, 'ms_per_step': 1e3*eager_s/args.steps, 'timed_regions': int(len(eager_runs)),
                  'step_ms_hip_events': {'min': float(step_ms.min()), 'median': float(np.median(step_ms)), 'max': float(step_ms.max())}},
        'roofline': {
            # (<IMPL, RW, OBS, SHADE, NG>; NG - 64-ray groups per wave - as ms_render picks it: four from 256 rays up on launches
            # of two and a half rounds of such waves)
            'kernel': 'ms_render = render_kernel<2,1,%s,%d> (headings cached by ms_physics)' % (
                '1,0' if args.depth_only else '0,1', ray_groups(dev, core)) + (' with STEP = 1: the env\'s physics step in the same wave'
                                                                                 if args.one_launch else ''),
            'bound': 'hbm', 'achieved': achieved,
            'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': achieved/HBM_PEAK_GBPS,
            'traffic': traffic, 'traffic_source': traffic_source,
            'algorithmic_bytes_per_launch': rb, 'avg_launch_ms': render_ms,
            'avg_launch_ms_source': 'median over HIP-event pairs on the stream around every ms_render of the eager leg (in place, '
                                    'between the physics launches; the pair also times its own two event packets, about 1 us)',
            'launch_ms_min_max': [float(render_each.min()), float(render_each.max())], 'launches_timed': int(len(render_each)),
            'step_algorithmic_bytes': rb + pb, 'step_achieved_GBps': (rb + pb)/(ms_per_step*1e-3)/1e9,
            # the raycast meets the fp32 VALU ceiling before the HBM one - the reference's all-pairs work
            # over this kernel's time, against the vector peak (the kernel culls, so this is an equivalent rate)
            'valu': {'all_pairs_flops_per_launch': flops, 'all_pairs_equivalent_TFLOPs': flops/(render_ms*1e-3)/1e12,
                     'peak_fp32_vector_TFLOPs': FP32_VECTOR_PEAK_TFLOPS,
                     'frac_of_peak': flops/(render_ms*1e-3)/1e12/FP32_VECTOR_PEAK_TFLOPS}},
    }
    if distributed:
        # what the driver needs to read a 1 -> 8 curve: whose time `value` is, and what each rank's host side ran with
        aff = sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else []
        mine = {'rank': rank, 'device': str(device), 'omp_num_threads': os.environ.get('OMP_NUM_THREADS'),
                'torch_threads': torch.get_num_threads(), 'cpus_allowed': len(aff)}
        hosts = [None]*world
        dist.all_gather_object(hosts, mine)
        out['per_rank']['host'] = hosts
        out['timing'] = ('value = all ranks\' envs x K / MAX over ranks of each rank\'s OWN synchronize-to-synchronize time for the K steps '
                         '(start barrier before the clock starts; nothing collective inside it), median over the timed regions')
    if args.share_gpu and distributed:
        out['data'] = 'synthetic; RANKS SHARE A GPU (--share-gpu: a plumbing check, not a measurement)'
    if args.baseline_line:
        base = [json.loads(l) for l in open(args.baseline_line) if l.lstrip().startswith('{')][-1]
        out['scaling_efficiency'] = {'vs': os.path.basename(args.baseline_line), 'n1_value': base['value'], 'n1_gpus': base.get('n_gpus', 1),
                                     'efficiency': value/(world*base['value']/max(base.get('n_gpus', 1), 1))}
    if extras:
        def leg(name, fn):
            """An auxiliary leg of the line: what it returns - or, if it fails, the reason in its place. The headline (`value`,
            `roofline`) has been measured by now and is printed whatever happens to the legs after it."""
            try:
                return fn()
            except Exception as e:                                         # noqa: BLE001 (reported, not swallowed: the line says what failed)
                import traceback
                log(f'{name} FAILED: {type(e).__name__}: {e}')
                torch.cuda.empty_cache()
                return {'error': f'{type(e).__name__}: {e}'[:400], 'where': traceback.format_exc(limit=3)[-600:]}
        if not args.no_cpu_baseline:
            got = leg('cpu baselines', lambda: cpu_baselines(core))
            out.update(got if 'error' not in got else {'cpu_baseline': {'value': None, 'unit': 'env-steps/s', 'cores': 0, 'kind': 'port',
                                                                        'sample': 'failed', **got}})
            log('cpu baselines done')
        if not args.no_env_fps:
            out['env_step_headline_shape'] = leg('env.step at the headline shape', lambda: headline_env_step(dev, core))
            log('env.step at the headline shape done')
        del core, scenery
        torch.cuda.empty_cache()
        if not args.no_shapes:
            out['shapes'] = leg('other shapes', lambda: other_shapes(dev))
            log('other shapes done')
        if not args.no_env_fps:
            out['env_step'] = leg('env-step rates', lambda: env_step_fps(device))
            log('env-step rates done')
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()
    return out


if __name__ == '__main__':
    main()
