"""The envs' own logic - who shoots whom, respawns, lifespans - against numpy restatements of the reference's formulas
on synthetic inputs. Pure tensor logic over the hot path's outputs, so it runs without a GPU."""
import numpy as np
import pytest
import torch


# ---- reference restatements (numpy) ---------------------------------------------------------------------------------

def ref_observe_opponents(indices, n_model, n_agents, subsample):
    """demo/envs/deathmatch.py:74-80 `_observe`: indices (F, A, 1, res) -> opponents (F, A, 1, res/subsample)."""
    F, A, _, res = indices.shape
    line_idxs = indices.reshape(F, A, 1, res//subsample, subsample)[..., subsample//2]
    obj_idxs = line_idxs//n_model
    mask = (0 <= line_idxs) & (obj_idxs < n_agents)
    return np.where(mask, obj_idxs, -1)


def ref_shoot(opponents, n_agents, positions, bounds, health, damage, clearance=1.):
    """demo/envs/deathmatch.py:54-72 `_shoot`. Returns (matchings, hits, new health, new damage)."""
    res = opponents.shape[-1]
    middle = slice(res//2 - 1, res//2 + 1)
    agents = np.arange(n_agents)
    matchings = (opponents[:, :, None] == agents[None, None, :, None, None])[..., middle].any(-1).any(-1)
    hits = matchings.sum(2).astype(np.float32)
    wounds = matchings.sum(1).astype(np.float32)
    damage = damage + np.float32(.05)*hits
    outside = (positions < -clearance).any(-1) | (positions > (bounds[:, None] + clearance)).any(-1)
    health = health + np.float32(-.05)*(wounds + outside) - np.float32(.001)
    return matchings, hits.reshape(-1), health, damage


# ---- Deathmatch -----------------------------------------------------------------------------------------------------

def _synthetic_indices(rng, F, A, M, res):
    idx = rng.randint(A*M, A*M + 300, (F, A, 1, res))                    # walls
    idx[rng.uniform(size=idx.shape) < .1] = -1                           # misses
    agent_hit = rng.uniform(size=idx.shape) < .35
    idx[agent_hit] = rng.randint(0, A*M, agent_hit.sum())                # agent lines, own included
    return idx.astype(np.int32)


@pytest.mark.parametrize('A,res,sub', [(4, 512, 4), (2, 64, 1), (3, 128, 8), (6, 32, 2)])
def test_crosshair_matrix_is_the_reference_matching(A, res, sub):
    from megastep_amd.demo.envs import deathmatch
    rng = np.random.RandomState(A*res + sub)
    M, F = 8, 50
    idx = _synthetic_indices(rng, F, A, M, res)
    got = deathmatch.crosshair_matrix(torch.as_tensor(idx), M, A, sub).numpy()
    opp = ref_observe_opponents(idx, M, A, sub)
    want = ref_shoot(opp, A, np.zeros((F, A, 2), np.float32), np.ones((F, 2), np.float32),
                     np.ones((F, A), np.float32), np.zeros((F, A), np.float32))[0]
    np.testing.assert_array_equal(got, want)
    assert want.any() and not want.all()


def test_exchange_fire_is_the_reference_shoot():
    """health, damage and reward after a frame: `_exchange_fire` against deathmatch.py:54-72 on synthetic hit lines,
    with some agents strayed outside their floorplan."""
    from megastep_amd import core, scene, toys, modules
    from megastep_amd.demo.envs import deathmatch
    rng = np.random.RandomState(0)
    F, A, res, sub = 20, 4, 512, 4
    sc = scene.scenery(F*[toys.box()], A, device='cpu', bake=False)
    env = deathmatch.Deathmatch.__new__(deathmatch.Deathmatch)          # the pieces _exchange_fire touches, no GPU
    env.core = core.Core(sc, res=res, fov=70)
    env._rgb = modules.RGB(env.core, n_agents=1, subsample=sub)
    bounds = rng.uniform(5, 10, (F, 2)).astype(np.float32)
    env._bounds = torch.as_tensor(bounds)
    env._upper = env._bounds[:, None] + deathmatch.CLEARANCE
    pos = rng.uniform(-2, 12, (F, A, 2)).astype(np.float32)
    env.core.agents.positions[:] = torch.as_tensor(pos)
    health, damage = rng.uniform(0, 1, (F, A)).astype(np.float32), rng.uniform(0, 1, (F, A)).astype(np.float32)
    env._health, env._damage = torch.as_tensor(health.copy()), torch.as_tensor(damage.copy())
    idx = _synthetic_indices(rng, F, A, 8, res)
    reward = env._exchange_fire(torch.as_tensor(idx)[..., :])
    opp = ref_observe_opponents(idx, 8, A, sub)
    matchings, hits, want_health, want_damage = ref_shoot(opp, A, pos, bounds, health, damage)
    np.testing.assert_array_equal(env.matchings.numpy(), matchings)
    np.testing.assert_array_equal(reward.numpy(), hits)
    np.testing.assert_allclose(env._health.numpy(), want_health, rtol=0, atol=1e-6)
    np.testing.assert_allclose(env._damage.numpy(), want_damage, rtol=0, atol=1e-6)
    assert ((pos < -1).any(-1) | (pos > bounds[:, None] + 1).any(-1)).any()


# ---- RandomSpawns ---------------------------------------------------------------------------------------------------

def test_random_spawns_touch_only_the_reset_agents():
    """modules.py:312-326: reset agents get a pose from THEIR row of the spawn table and zero velocities; everyone else
    is left exactly as they were. The reference draws the spawn index below ``spawns.angles.shape[1]`` (modules.py:321)
    - which is the number of AGENTS, not of spawn points - so the choice is uniform over the first ``n_agents`` entries
    of the table; a drop-in keeps that."""
    from megastep_amd import core, scene, cubicasa, modules
    np.random.seed(0); torch.manual_seed(0)
    geoms = cubicasa.sample(6, n_unique=16)
    c = core.Core(scene.scenery(geoms, 3, device='cpu', bake=False))
    spawner = modules.RandomSpawns(geoms, c, n_spawns=20)
    table_p, table_a = spawner._spawns.positions.numpy(), spawner._spawns.angles.numpy()
    rng = np.random.RandomState(1)
    counts = np.zeros(20)
    for trial in range(200):
        before = {k: torch.as_tensor(rng.normal(size=getattr(c.agents, k).shape).astype(np.float32)) for k in
                  ('angles', 'positions', 'velocity', 'angvelocity')}
        for k, v in before.items():
            getattr(c.agents, k)[:] = v
        reset = torch.as_tensor(rng.uniform(size=(6, 3)) < .4)
        spawner(reset)
        m = reset.numpy()
        for k, v in before.items():
            np.testing.assert_array_equal(getattr(c.agents, k).numpy()[~m], v.numpy()[~m], err_msg=k)   # untouched
        assert (c.agents.velocity.numpy()[m] == 0).all() and (c.agents.angvelocity.numpy()[m] == 0).all()
        for e, a in zip(*np.nonzero(m)):
            which = np.flatnonzero((table_p[e, a] == c.agents.positions.numpy()[e, a]).all(-1)
                                   & (table_a[e, a] == c.agents.angles.numpy()[e, a]))
            assert len(which) >= 1, 'pose is not from this agent\'s spawn table'
            counts[which[0]] += 1
    assert (counts[3:] == 0).all() and counts[:3].min() > .8*counts[:3].mean()      # uniform over the first n_agents = 3


def test_random_spawn_tables_hold_free_cells_of_their_own_geometry():
    from megastep_amd import core, scene, cubicasa, modules, geometry
    np.random.seed(0); torch.manual_seed(0)
    geoms = cubicasa.sample(4, n_unique=16)
    geoms = [geoms[0], geoms[1], geoms[0], geoms[2], geoms[3]]
    c = core.Core(scene.scenery(geoms, 2, device='cpu', bake=False))
    for fast in (False, True):
        spawner = modules.RandomSpawns(geoms, c, n_spawns=50, fast=fast)
        assert spawner._spawns.positions.shape == (5, 2, 50, 2) and spawner._spawns.angles.shape == (5, 2, 50)
        assert spawner._spawns.angles.abs().max() <= 180
        for e, g in enumerate(geoms):
            pts = spawner._spawns.positions[e].reshape(-1, 2).numpy().astype(float)
            ij = geometry.indices(pts, g['masks'].shape, g['res'])
            assert (g['masks'][ij[:, 0], ij[:, 1]] > 0).all(), (fast, e)


# ---- RandomLifespans ------------------------------------------------------------------------------------------------

def test_random_lifespans_follow_the_reference_rules():
    """modules.py:361-366: every call ages everyone by one step; those that reached their maximum - or are reset from
    outside - are flagged, start again at zero and get a fresh maximum in [min, max)."""
    from megastep_amd import core, scene, toys, modules
    torch.manual_seed(0)
    c = core.Core(scene.scenery(32*[toys.box()], 3, device='cpu', bake=False))
    life = modules.RandomLifespans(c, max_lifespan=12)                   # min defaults to 6
    assert life.min_lifespan == 6 and life.max_lifespan == 12
    age, limit = np.zeros((32, 3), np.int64), life._max_lifespans.numpy().copy()
    assert ((limit >= 6) & (limit < 12)).all() and len(np.unique(limit)) > 3
    rng = np.random.RandomState(0)
    flagged = 0
    for step in range(60):
        outside = torch.as_tensor(rng.uniform(size=(32, 3)) < .05) if step % 3 == 0 else None
        got = life(outside).numpy()
        age += 1
        want = (age >= limit) | (outside.numpy() if outside is not None else False)
        np.testing.assert_array_equal(got, want)
        age[want] = 0
        np.testing.assert_array_equal(life._lifespans.numpy(), age)
        fresh = life._max_lifespans.numpy()
        np.testing.assert_array_equal(fresh[~want], limit[~want])       # maxima only change on a reset
        assert ((fresh >= 6) & (fresh < 12)).all()
        limit = fresh.copy()
        flagged += want.sum()
    assert flagged > 100
    st = life.state(3)
    assert st.lifespan.shape == (3,) and st.max_lifespans.shape == (3,)


# ---- Explorer's books ----------------------------------------------------------------------------------------------

def test_seen_texel_books_follow_the_reference_bookkeeping():
    """explorer.py:45-58 by other means: the reference re-counts the seen texels of every env each step and rewards the
    difference; `SeenTexels` keeps a tally the render kernel adds to, a copy of it from the last call, and an epoch per
    env that a respawn bumps (forgetting every texel at once). Here the kernel's part is played by hand: stamping texels
    and raising the tally. The third row of `counters` is the env's own (Explorer keeps episode lengths there) and is
    cleared with the rest."""
    from megastep_amd import scene, toys
    from megastep_amd.demo.envs import explorer
    F = 5
    sc = scene.scenery(F*[toys.box()], 1, device='cpu', bake=False)
    books = explorer.SeenTexels(sc, F)
    T = len(books.texel_env)
    assert books.counters.shape == (3, F) and books.stamp.shape == (T,) and books.mask().sum() == 0
    per_env = T//F
    rng = np.random.RandomState(0)
    seen = np.zeros((F, per_env), bool)                        # the reference's `_seen`, env by env
    potential = np.zeros(F)
    books.spare += 7
    for step in range(12):
        look = rng.uniform(size=(F, per_env)) < .1            # texels under this frame's rays
        respawn = rng.uniform(size=F) < (.25 if step % 3 == 2 else 0.)
        # reference order (explorer.py:83-95): respawned envs forget, then the frame is looked at
        books.forget(torch.as_tensor(respawn))
        seen[respawn] = False
        potential[respawn] = 0
        # the kernel: stamp what is in view with the env's epoch, add the texels that did not carry it to the tally
        stamp = books.stamp.view(F, per_env).numpy()
        epoch = books.epoch.numpy()
        fresh = look & (stamp != epoch[:, None])
        stamp[look] = np.broadcast_to(epoch[:, None], look.shape)[look]
        books.tally += torch.as_tensor(fresh.sum(1).astype(np.int32))
        # the reference: potential = seen texels per env, reward = its increase
        seen |= look
        new_potential = seen.sum(1)
        np.testing.assert_array_equal(books.gained().numpy(), new_potential - potential)
        potential = new_potential
        np.testing.assert_array_equal(books.count.numpy(), potential)
        np.testing.assert_array_equal(books.mask().view(F, per_env).numpy(), seen)
        assert (books.spare.numpy()[respawn] == 0).all() and (books.gained() == 0).all()
    assert potential.max() > 10


def test_an_imu_reading_taken_inside_the_physics_launch_goes_stale_with_the_agents():
    """modules.IMU hands out the reading the physics launch took (`_pending`) only while nobody has touched the agents
    since: a respawn through the modules' helpers, or another physics call, moves the agents' epoch on and the reading
    is worked out afresh (round 2's advisor: a reset between mover(..., imu=) and imu() returned the old observation)."""
    from megastep_amd import core, modules, scene, toys
    sc = scene.scenery(3*[toys.box()], 2, device='cpu', bake=False)
    c = core.Core(sc, res=8)
    c.agents.velocity[:] = torch.tensor([1., 2.])
    c.agents.angvelocity[:] = 90.
    imu = modules.IMU(c)
    fresh = imu().clone()
    stale = torch.full_like(fresh, 7.)
    imu._pending = (stale, c.agents._epoch)                              # as _move leaves it after a fused physics call
    assert torch.equal(imu(), stale) and imu._pending is None
    imu._pending = (stale, c.agents._epoch)
    spawns = modules.RandomSpawns(3*[toys.box()], c)
    spawns(c.agent_full(True))                                           # a respawn with tensor ops: velocities are zero now
    assert c.agents._epoch > 0 and torch.equal(imu(), torch.zeros_like(fresh))


# ---- the same, against the REFERENCE's own methods (tests/golden/make_golden.py: deathmatch.py:54-80 `_observe` + `_shoot`,
# ---- explorer.py:34-58 `_tex_indices` + `_reward`, called unbound on stand-ins holding seeded tensors) ----------------------

def _golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_host.npz'))


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_deathmatch_frame_equals_the_references_observe_and_shoot(tag):
    """`crosshair_matrix` / `_exchange_fire` on the inputs the reference's `_observe` + `_shoot` were run on: the same
    matchings, hits (the reward), health and damage. Also with the crosshair ids the render kernel writes (the two centre
    pixels' agents) standing in for the full plane of hit lines."""
    from megastep_amd import core, scene, toys, modules
    from megastep_amd.demo.envs import deathmatch
    g = _golden()
    F, A, res, sub, M = (int(v) for v in g[f'dm_{tag}_shape'])
    idx = torch.as_tensor(g[f'dm_{tag}_indices'])
    np.testing.assert_array_equal(deathmatch.crosshair_matrix(idx, M, A, sub).numpy(), g[f'dm_{tag}_matchings'])
    for from_centre in (False, True):
        sc = scene.scenery(F*[toys.box()], A, device='cpu', bake=False)
        assert sc.model.shape[0] == M
        env = deathmatch.Deathmatch.__new__(deathmatch.Deathmatch)
        env.core = core.Core(sc, res=res, fov=70)
        env._rgb = modules.RGB(env.core, n_agents=1, subsample=sub)
        env._bounds = torch.as_tensor(g[f'dm_{tag}_bounds'])
        env._upper = env._bounds[:, None] + deathmatch.CLEARANCE
        env._everyone = torch.arange(A)
        env.core.agents.positions[:] = torch.as_tensor(g[f'dm_{tag}_positions'])
        env._health, env._damage = torch.as_tensor(g[f'dm_{tag}_health0'].copy()), torch.as_tensor(g[f'dm_{tag}_damage0'].copy())
        if from_centre:
            # what render_kernel's obs_centre holds: for the two central observation pixels the agent their middle ray landed on
            W = res//sub
            mid = idx[:, :, 0, [(W//2 - 1)*sub + sub//2, (W//2)*sub + sub//2]]
            centre = torch.where((mid >= 0) & (mid < A*M), torch.div(mid, M, rounding_mode='floor'), torch.full_like(mid, -1))
            reward = env._exchange_fire(centre=centre)
        else:
            reward = env._exchange_fire(idx)
        np.testing.assert_array_equal(env.matchings.numpy(), g[f'dm_{tag}_matchings'])
        np.testing.assert_array_equal(reward.numpy(), g[f'dm_{tag}_hits'])
        np.testing.assert_allclose(env._health.numpy(), g[f'dm_{tag}_health'], rtol=0, atol=1e-6)
        np.testing.assert_allclose(env._damage.numpy(), g[f'dm_{tag}_damage'], rtol=0, atol=1e-6)
        np.testing.assert_allclose(env._health.unsqueeze(-1).numpy(), g[f'dm_{tag}_obs_health'], rtol=0, atol=1e-6)
    assert g[f'dm_{tag}_matchings'].any() and (g[f'dm_{tag}_hits'] > 0).any()


def test_explorer_books_equal_the_references_tex_indices_and_reward():
    """Ten frames the reference's `_tex_indices` + `_reward` were run on (respawns in two of them; from the seventh on
    some rays miss): `texels_hit` names the same texel under every ray, and `SeenTexels` - the render kernel's part
    played by hand, as it is written in render_kernel: stamp the texel under every ray with its env's epoch, count the
    ones that did not carry it; a ray that MISSED stamps the scenery's last texel to the credit of the last env, which is
    what the reference's `_seen[-1] = True` amounts to (explorer.py:36,47) - hands out the same rewards, potentials and
    seen masks."""
    from megastep_amd import dotdict
    from megastep_amd.demo.envs import explorer
    g = _golden()
    N, R, sub, T, frames = (int(v) for v in g['ex_shape'])
    tex_w, tex_s = torch.as_tensor(g['ex_tex_widths']), torch.as_tensor(g['ex_tex_starts'])
    line_of_texel = torch.repeat_interleave(torch.arange(len(tex_w)), tex_w.long())
    line_env = torch.as_tensor(np.searchsorted(g['ex_line_starts'], np.arange(len(tex_w)), side='right') - 1)
    sc = dotdict.dotdict(lines=dotdict.dotdict(starts=torch.as_tensor(g['ex_line_starts'].astype(np.int32)), inverse=line_env.int()),
                         textures=dotdict.dotdict(widths=tex_w, starts=tex_s, inverse=line_of_texel.int(), vals=torch.zeros((T, 3))))
    books = explorer.SeenTexels(sc, N)
    np.testing.assert_array_equal(books.texel_env.numpy(), g['ex_tex_to_env'])
    assert (g['ex_indices'] < 0).any() and g['ex_resets'].any()
    for f in range(frames):
        frame = dotdict.dotdict(indices=torch.as_tensor(g['ex_indices'][f]), locations=torch.as_tensor(g['ex_locations'][f]))
        texels = explorer.texels_hit(sc, frame)
        np.testing.assert_array_equal(texels.unsqueeze(2).numpy(), g['ex_tex_indices'][f])
        reset = torch.as_tensor(g['ex_resets'][f])
        books.forget(reset)
        # the kernel's part (render_kernel, `seen_stamp`)
        env_of_ray = torch.arange(N)[:, None, None, None].expand_as(texels)
        hit = texels >= 0
        t = torch.where(hit, texels, torch.full_like(texels, T - 1))
        e = torch.where(hit, env_of_ray, torch.full_like(env_of_ray, N - 1))
        fresh = torch.zeros(T, dtype=torch.bool)
        fresh[t.flatten()] = books.stamp[t.flatten()] != books.epoch[e.flatten()]
        books.stamp[t.flatten()] = books.epoch[e.flatten()]
        books.tally += torch.zeros(N, dtype=torch.int32).scatter_add_(0, books.texel_env[fresh], torch.ones(int(fresh.sum()), dtype=torch.int32))
        reward = (books.gained()/(R//sub)).masked_fill_(reset, 0.)
        np.testing.assert_allclose(reward.numpy(), g['ex_rewards'][f], rtol=0, atol=1e-6)
        np.testing.assert_array_equal(books.count.numpy(), g['ex_potentials'][f])
        np.testing.assert_array_equal(books.mask().numpy(), g['ex_seen'][f].astype(bool))
    assert g['ex_seen'][-1][-1] and g['ex_potentials'][-1].min() > 10
