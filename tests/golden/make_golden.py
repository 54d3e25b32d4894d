"""Generates tests/golden/reference_host.npz by importing the REFERENCE's Python host modules in this container.

Run from the repo root:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's package __init__ JIT-compiles a CUDA extension (impossible here) and some modules import packages this
image lacks (rasterio, shapely, bs4; matplotlib.tight_bbox is gone from mpl 3.10), so those names are stubbed in
sys.modules before import; everything recorded below is computed by the reference's own, unmodified Python code.
Only inputs and outputs are stored - no reference source. The kernels (physics/render/bake) cannot be run here at all:
they are pinned by the docs' known answer and analytic cases in tests/test_oracle.py instead.
"""
import os
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_host.npz')


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    pkg = _stub('megastep')
    pkg.__path__ = [os.path.join(REF, 'megastep')]
    calls = []

    class FakePhysics:
        progress = None

    cuda = _stub('megastep.cuda', physics=lambda scenery, agents: calls.append('physics') or FakePhysics())
    pkg.cuda = cuda
    pkg.plotting = _stub('megastep.plotting')
    _stub('rasterio'); _stub('rasterio.features'); sys.modules['rasterio'].features = sys.modules['rasterio.features']
    _stub('shapely'); _stub('shapely.ops', cascaded_union=None); _stub('shapely.geometry', Polygon=None, LineString=None)
    _stub('bs4', BeautifulSoup=None)
    import importlib
    mods = {k: importlib.import_module(f'megastep.{k}') for k in ('core', 'ragged', 'scene', 'geometry', 'modules', 'spaces')}
    mods['arrdict'] = importlib.import_module('rebar.arrdict')
    mods['dotdict'] = importlib.import_module('rebar.dotdict')
    # the demo envs (their package __init__ pulls in the RL stack: a shell instead, the two env modules from their files)
    pkg.cubicasa = _stub('megastep.cubicasa', sample=None)
    demo = _stub('megastep.demo'); demo.__path__ = [os.path.join(REF, 'megastep', 'demo')]
    envs = _stub('megastep.demo.envs'); envs.__path__ = [os.path.join(REF, 'megastep', 'demo', 'envs')]
    try:
        import matplotlib
        matplotlib.use('Agg')
    except ImportError:
        _stub('matplotlib'); _stub('matplotlib.pyplot')
    for k in ('deathmatch', 'explorer'):
        mods[k] = importlib.import_module(f'megastep.demo.envs.{k}')
    return mods, calls


def env_glue(m, g):
    """The demo envs' own arithmetic (deathmatch.py:54-80 `_observe` + `_shoot`, explorer.py:34-58 `_tex_indices` +
    `_reward`), run UNBOUND on stand-ins for `self` that hold seeded tensors: what the hot path's outputs are turned
    into. Inputs and outputs only."""
    import torch
    arrdict, dotdict = m['arrdict'], m['dotdict']
    Deathmatch, Explorer = m['deathmatch'].Deathmatch, m['explorer'].Explorer
    rng = np.random.RandomState(17)

    # ---- Deathmatch: who is in whose crosshair, hits, wounds, strays
    for tag, (F, A, res, sub) in {'a': (12, 4, 512, 4), 'b': (9, 3, 64, 2)}.items():
        M = 8
        idx = rng.randint(A*M, A*M + 300, (F, A, 1, res))                          # walls ...
        idx[rng.uniform(size=idx.shape) < .1] = -1                                 # ... misses ...
        on_agent = rng.uniform(size=idx.shape) < .35
        idx[on_agent] = rng.randint(0, A*M, on_agent.sum())                        # ... and agents' lines, own included
        idx = idx.astype(np.int32)
        pos = rng.uniform(-2, 12, (F, A, 2)).astype(np.float32)
        bounds = rng.uniform(5, 10, (F, 2)).astype(np.float32)
        health, damage = rng.uniform(0, 1, (F, A)).astype(np.float32), rng.uniform(0, 1, (F, A)).astype(np.float32)

        class Obs:                                                                  # RGB / Depth / IMU stand-ins
            subsample = sub
            def __call__(self, r=None):
                return torch.zeros(1)
        fake = type('FakeDeathmatch', (), {})()
        fake.core = dotdict.dotdict(n_agents=A, device='cpu', scenery=dotdict.dotdict(model=torch.zeros((M, 2, 2))),
                                    agents=dotdict.dotdict(positions=torch.as_tensor(pos)))
        fake._rgb = fake._depth = fake._imu = Obs()
        fake._bounds = torch.as_tensor(bounds)
        fake._health, fake._damage = torch.as_tensor(health.copy()), torch.as_tensor(damage.copy())
        fake._shoot = lambda opponents, fake=fake: Deathmatch._shoot(fake, opponents)
        frame = arrdict.arrdict(indices=torch.as_tensor(idx))
        m['deathmatch'].modules.render, real = (lambda core: frame), m['deathmatch'].modules.render
        obs, hits = Deathmatch._observe(fake)
        m['deathmatch'].modules.render = real
        g[f'dm_{tag}_shape'] = np.array([F, A, res, sub, M])
        g[f'dm_{tag}_indices'], g[f'dm_{tag}_positions'], g[f'dm_{tag}_bounds'] = idx, pos, bounds
        g[f'dm_{tag}_health0'], g[f'dm_{tag}_damage0'] = health, damage
        g[f'dm_{tag}_matchings'], g[f'dm_{tag}_hits'] = fake.matchings.numpy(), hits.numpy()
        g[f'dm_{tag}_health'], g[f'dm_{tag}_damage'] = fake._health.numpy(), fake._damage.numpy()
        g[f'dm_{tag}_obs_health'] = obs.health.numpy()

    # ---- Explorer: the texel under each ray, what has been seen, the reward for seeing it first
    N, R, sub = 5, 32, 4
    n_lines = rng.randint(9, 14, N)
    line_starts = np.cumsum(n_lines) - n_lines
    tex_w = rng.randint(1, 9, n_lines.sum()).astype(np.int32)
    tex_s = (np.cumsum(tex_w) - tex_w).astype(np.int32)
    T = int(tex_w.sum())
    tex_to_env = np.repeat(np.repeat(np.arange(N), n_lines), tex_w)
    fake = type('FakeExplorer', (), {})()
    fake.core = dotdict.dotdict(res=R, scenery=dotdict.dotdict(
        lines=dotdict.dotdict(starts=torch.as_tensor(line_starts.astype(np.int32))),
        textures=dotdict.dotdict(widths=torch.as_tensor(tex_w), starts=torch.as_tensor(tex_s))))
    fake._rgb = dotdict.dotdict(subsample=sub)
    fake._tex_to_env = torch.as_tensor(tex_to_env).long()
    fake._seen = torch.full_like(fake._tex_to_env, False)
    fake._potential = torch.zeros(N)
    fake._tex_indices = lambda aux, fake=fake: Explorer._tex_indices(fake, aux)
    frames = 10
    g['ex_shape'] = np.array([N, R, sub, T, frames])
    g['ex_line_starts'], g['ex_tex_widths'], g['ex_tex_starts'], g['ex_tex_to_env'] = line_starts, tex_w, tex_s, tex_to_env
    idxs, locs, resets, tis, rewards, potentials, seens = [], [], [], [], [], [], []
    for f in range(frames):
        idx = np.stack([rng.randint(0, n_lines[e], (1, 1, R)) for e in range(N)]).astype(np.int32)
        if f >= 6:
            idx[rng.uniform(size=idx.shape) < .08] = -1              # (from frame 6 on some rays miss: explorer.py:36,47 index _seen[-1])
        loc = rng.uniform(0, 1, idx.shape).astype(np.float32)
        loc[rng.uniform(size=loc.shape) < .05] = 1.                 # the far end of a line
        loc[idx < 0] = np.nan
        reset = rng.uniform(size=N) < (.3 if f in (3, 7) else 0.)
        # the order of Explorer.step (explorer.py:83-95): _reset forgets, then _observe looks and rewards
        rt = torch.as_tensor(reset)
        fake._seen[rt[fake._tex_to_env]] = False
        fake._potential[rt] = 0
        aux = arrdict.arrdict(indices=torch.as_tensor(idx), locations=torch.as_tensor(loc))
        tis.append(Explorer._tex_indices(fake, aux).numpy())
        reward = Explorer._reward(fake, aux, rt)
        idxs.append(idx); locs.append(loc); resets.append(reset)
        rewards.append(reward.numpy().copy()); potentials.append(fake._potential.numpy().copy()); seens.append(fake._seen.numpy().copy())
    for k, v in dict(indices=idxs, locations=locs, resets=resets, tex_indices=tis, rewards=rewards, potentials=potentials, seen=seens).items():
        g[f'ex_{k}'] = np.stack(v)


def main():
    import torch
    m, calls = import_reference()
    scene, geometry, modules, core, ragged, arrdict = (m[k] for k in ('scene', 'geometry', 'modules', 'core', 'ragged', 'arrdict'))
    g = {}

    # --- constants (core.py:10-14)
    g['AGENT_WIDTH'], g['TEXTURE_RES'], g['AGENT_RADIUS'] = core.AGENT_WIDTH, core.TEXTURE_RES, core.AGENT_RADIUS
    g['gamma_decode_in'] = np.linspace(0, 1, 7)
    g['gamma_decode_out'] = core.gamma_decode(g['gamma_decode_in'])
    g['gamma_encode_out'] = core.gamma_encode(g['gamma_decode_in'])

    # --- scene.py
    g['agent_model'] = scene.agent_model()
    g['agent_colors'] = scene.agent_colors()
    # toys.box(5) walls, as the reference's toys.py:5-16 computes them (its masks() needs rasterio, so by hand)
    corners = [(np.cos(t), np.sin(t)) for t in np.arange(np.pi/4, 2*np.pi, np.pi/2)]
    corners = 5/2**.5*np.array(corners) + 5/2 + geometry.MARGIN
    box_walls = np.stack(geometry.cyclic_pairs(corners))
    g['box_walls'] = box_walls
    rng = np.random.RandomState(3)
    walls = rng.uniform(1, 9, (23, 2, 2))
    g['walls'] = walls
    g['lengths'] = scene.lengths(walls)
    g['resolutions'] = scene.resolutions(np.concatenate([scene.agent_model(), walls]))
    g['wall_pattern'] = scene.wall_pattern(500, random=np.random.RandomState(5))
    for n_agents in (1, 3):
        agentlines = np.tile(scene.agent_model(), (n_agents, 1, 1))
        agentcolors = np.tile(scene.agent_colors(), (n_agents, 1))
        tex, widths = scene.init_textures(agentlines, agentcolors, box_walls, np.random.RandomState(1))
        g[f'box_textures_{n_agents}'], g[f'box_texwidths_{n_agents}'] = tex, widths
        tex, widths = scene.init_textures(agentlines, agentcolors, walls, np.random.RandomState(2))
        g[f'walls_textures_{n_agents}'], g[f'walls_texwidths_{n_agents}'] = tex, widths
    lights = rng.uniform(1, 9, (5, 2))
    g['lights'] = lights
    g['random_lights'] = scene.random_lights(lights, random=np.random.RandomState(11))

    # --- ragged.py (RaggedNumpy; the reference's own test vector ragged.py:93-103 plus a bigger one)
    for name, vals, widths in [('small', np.array([0, 1, 2, 3, 4, 5]), np.array([3, 1, 2])),
                               ('big', rng.uniform(size=(40, 3)), np.array([5, 1, 9, 2, 7, 16]))]:
        r = ragged.RaggedNumpy(vals, widths)
        g[f'ragged_{name}_vals'], g[f'ragged_{name}_widths'] = vals, widths
        g[f'ragged_{name}_starts'], g[f'ragged_{name}_ends'], g[f'ragged_{name}_inverse'] = r.starts, r.ends, r.inverse
        g[f'ragged_{name}_item1'], g[f'ragged_{name}_itemlast'] = r[1], r[-1]
        g[f'ragged_{name}_slice_vals'], g[f'ragged_{name}_slice_widths'] = r[1:3].vals, r[1:3].widths

    # --- geometry.py
    idx = rng.randint(0, 30, (7, 4, 2))
    g['centers_in'], g['centers_out'] = idx, geometry.centers(idx, (36, 41), .2)
    xy = rng.uniform(-1, 9, (9, 2))
    g['indices_in'], g['indices_out'] = xy, geometry.indices(xy, (36, 41), .2)
    g['centers_origin'] = geometry.centers(np.array([[0, 0]]), (36, 36), .2)
    dup = np.concatenate([walls[:6], walls[2:4], walls[4:5, ::-1], walls[6:9]])
    g['unique_in'], g['unique_out'] = dup, geometry.unique(dup)
    g['signed_area'] = np.array([geometry.signed_area(corners), geometry.signed_area(corners[::-1])])

    # --- modules.py: frames
    ang = torch.tensor([[0., 90.], [-45., 170.]])
    vec = torch.tensor([[[1., 0.], [1., 0.]], [[.3, -2.], [0., 1.]]])
    g['frame_angles'], g['frame_vectors'] = ang.numpy(), vec.numpy()
    g['to_global'], g['to_local'] = modules.to_global_frame(ang, vec).numpy(), modules.to_local_frame(ang, vec).numpy()

    # --- modules.py: observation modules on a synthetic render
    class FakeCore:
        n_envs, n_agents, res, fps, device, agent_radius = 3, 2, 16, 10, 'cpu', core.AGENT_RADIUS
        random = np.random.RandomState(1)
    fc = FakeCore()
    tg = torch.Generator().manual_seed(0)
    dist = torch.rand((3, 2, 16), generator=tg)*12
    dist[0, 0, 3] = float('inf')
    screen = torch.rand((3, 2, 16, 3), generator=tg)
    r = arrdict.arrdict(distances=dist.unsqueeze(2), screen=screen.unsqueeze(2).permute(0, 1, 4, 2, 3))
    g['obs_distances'], g['obs_screen'] = dist.numpy(), screen.numpy()
    for sub in (1, 4):
        g[f'depth_sub{sub}'] = modules.Depth(fc, subsample=sub, max_depth=10)(r).numpy()
        g[f'rgb_sub{sub}'] = modules.RGB(fc, subsample=sub)(r).numpy()
    g['downsample'] = modules.downsample(dist, 4).numpy()

    # --- modules.py: movement (velocity tables and the state they write), physics stubbed out
    agents = arrdict.arrdict(
        angles=torch.tensor([[0., 90.], [-45., 170.], [10., -100.]]),
        positions=torch.zeros(3, 2, 2),
        angvelocity=torch.tensor([[0., 5.], [-3., 2.], [1., 1.]]),
        velocity=torch.tensor([[[1., 0.], [0., 1.]], [[.5, .5], [-1., 0.]], [[0., 0.], [2., -2.]]]))
    fc.agents, fc.scenery = agents, None
    actions = torch.tensor([[0, 1], [3, 5], [6, 2]])
    g['move_angles'], g['move_actions'] = agents.angles.numpy().copy(), actions.numpy()
    g['move_vel0'], g['move_angvel0'] = agents.velocity.numpy().copy(), agents.angvelocity.numpy().copy()
    mom = modules.MomentumMovement(fc, accel=5, ang_accel=180, decay=.125)
    mom(arrdict.arrdict(actions=actions))
    g['momentum_vel'], g['momentum_angvel'] = agents.velocity.numpy().copy(), agents.angvelocity.numpy().copy()
    g['momentum_actionset_v'], g['momentum_actionset_w'] = mom._actionset.velocity.numpy(), mom._actionset.angvelocity.numpy()
    simple = modules.SimpleMovement(fc, speed=10, ang_speed=180)
    simple(arrdict.arrdict(actions=actions))
    g['simple_vel'], g['simple_angvel'] = agents.velocity.numpy().copy(), agents.angvelocity.numpy().copy()
    assert calls == ['physics', 'physics']
    g['imu'] = modules.IMU(fc)().numpy()

    # --- modules.py: spawn tables from a hand-made mask (global np.random, seeded)
    mask = np.zeros((12, 15), dtype=np.int16)
    mask[2:9, 3:11] = 1
    mask[4, :] = -1
    geoms = [arrdict.arrdict(masks=mask, res=.2), arrdict.arrdict(masks=mask.T.copy(), res=.2)]
    np.random.seed(4)
    g['spawn_mask'] = mask
    g['spawn_positions'] = modules.random_empty_positions(geoms, 2, 10)
    fc.n_envs, fc.random = 2, np.random.RandomState(1)
    np.random.seed(4)
    sp = modules.RandomSpawns(geoms, fc, n_spawns=10)
    g['spawns_angles'], g['spawns_positions'] = sp._spawns.angles.numpy(), sp._spawns.positions.numpy()

    env_glue(m, g)

    np.savez_compressed(OUT, **{k: np.asarray(v) for k, v in g.items()})
    print(f'wrote {OUT}: {len(g)} arrays, {os.path.getsize(OUT)} bytes')


if __name__ == '__main__':
    main()
