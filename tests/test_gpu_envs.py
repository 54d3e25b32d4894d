"""The example environments (the reference's callers of the hot path, demo/envs/*.py) stepping on the GPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _decision(env, n):
    from megastep_amd import arrdict
    return arrdict.arrdict(actions=torch.randint(0, 7, (n, env.action_space.shape[0]), device='cuda'))


def test_minimal_env_shapes_and_motion():
    """docs/tutorials/minimal-env/index.rst:288-290: Minimal(128).reset().obs; code gives (N, 1, 3, 1, 64)."""
    from megastep_amd.demo import Minimal
    torch.manual_seed(0); np.random.seed(0)
    env = Minimal(128)
    world = env.reset()
    assert world.obs.shape == (128, 1, 3, 1, 64) and world.obs.dtype == torch.float32
    assert 0 <= world.obs.min() and world.obs.max() <= 1 and world.obs.max() > 0
    p0 = env.core.agents.positions.clone()
    for _ in range(5):
        world = env.step(_decision(env, 128))
    assert (env.core.agents.positions - p0).abs().max() > 0
    pos = env.core.agents.positions
    assert (pos > 1).all() and (pos < 6).all(), 'agents must stay inside the box'
    state = env.state(0)
    assert state.rgb.shape == (1, 3, 1, 64) and state.core.scenery.lines.shape == (12, 2, 2)


def test_explorer_env():
    from megastep_amd.demo import Explorer
    from megastep_amd import cubicasa
    torch.manual_seed(0); np.random.seed(0)
    env = Explorer(16, geometries=cubicasa.sample(16, n_unique=16))
    world = env.reset()
    assert world.obs.rgb.shape == (16, 1, 3, 1, 64) and world.obs.d.shape == (16, 1, 1, 1, 64) and world.obs.imu.shape == (16, 1, 3)
    assert world.reset.all() and (world.reward == 0).all()
    total = torch.zeros(16, device='cuda')
    for _ in range(20):
        world = env.step(_decision(env, 16))
        total += world.reward
    assert (total > 0).any() and torch.isfinite(total).all()          # new texels get seen
    assert (world.obs.d >= 0).all() and (world.obs.d <= 1).all()
    assert env.state(0).seen.dtype == torch.bool or env.state(0).seen.dtype == torch.int64


def test_deathmatch_env():
    from megastep_amd.demo import Deathmatch
    from megastep_amd import cubicasa
    torch.manual_seed(0); np.random.seed(0)
    env = Deathmatch(32, 4, geometries=cubicasa.sample(8, n_unique=16))
    assert env.n_envs == 32 and env.core.n_envs == 8 and env.core.res == 512
    world = env.reset()
    assert world.obs.rgb.shape == (32, 1, 3, 1, 128) and world.obs.d.shape == (32, 1, 1, 1, 128)
    assert world.obs.imu.shape == (32, 1, 3) and world.obs.health.shape == (32, 1, 1)
    assert world.reset.shape == (32,) and world.reward.shape == (32,)
    for _ in range(10):
        world = env.step(_decision(env, 32))
    assert torch.isfinite(world.obs.rgb).all() and torch.isfinite(world.obs.d).all()
    assert (env._health <= 1).all()
    assert env.state(0).matchings.shape == (4, 4)


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_deathmatch_logic_kernel_equals_the_references_observe_and_shoot(tag):
    """ms_deathmatch_shoot (the Deathmatch env's game logic as one launch) on the inputs the REFERENCE's `_observe` + `_shoot`
    were run on (tests/golden/make_golden.py: deathmatch.py:54-80 called unbound on seeded tensors): the same matchings, hits
    (the reward), health, damage and health observation; then the `_reset` half (deathmatch.py:46-52): agents marked dead start
    the frame from health 1 / damage 0, and `dead` comes back as health <= 0."""
    import os
    from megastep_amd import cuda
    from megastep_amd.demo.envs import deathmatch
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_host.npz'))
    F, A, res, sub, M = (int(v) for v in g[f'dm_{tag}_shape'])
    idx = torch.as_tensor(g[f'dm_{tag}_indices']).cuda()
    W = res//sub
    mid = idx[:, :, 0, [(W//2 - 1)*sub + sub//2, (W//2)*sub + sub//2]]       # what render_kernel's obs_centre holds
    centre = torch.where((mid >= 0) & (mid < A*M), torch.div(mid, M, rounding_mode='floor'), torch.full_like(mid, -1)).int().contiguous()
    positions = torch.as_tensor(g[f'dm_{tag}_positions']).cuda().contiguous()
    upper = (torch.as_tensor(g[f'dm_{tag}_bounds']).cuda() + deathmatch.CLEARANCE).float().contiguous()
    health, damage = torch.as_tensor(g[f'dm_{tag}_health0'].copy()).cuda(), torch.as_tensor(g[f'dm_{tag}_damage0'].copy()).cuda()
    dead = torch.zeros((F, A), dtype=torch.bool, device='cuda')
    reset, reward, health_obs, matchings = cuda.deathmatch_shoot(centre, positions, upper, health, damage, dead, matchings=True)
    np.testing.assert_array_equal(matchings.cpu().numpy(), g[f'dm_{tag}_matchings'])
    np.testing.assert_array_equal(reward.reshape(-1).cpu().numpy(), g[f'dm_{tag}_hits'])
    np.testing.assert_allclose(health.cpu().numpy(), g[f'dm_{tag}_health'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(damage.cpu().numpy(), g[f'dm_{tag}_damage'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(health_obs.unsqueeze(-1).cpu().numpy(), g[f'dm_{tag}_obs_health'], rtol=0, atol=1e-6)
    # bit for bit the reference's statements in binary32 (deathmatch.py:64,70)
    m = torch.as_tensor(g[f'dm_{tag}_matchings']).cuda()
    pos = positions
    outside = (pos < -deathmatch.CLEARANCE).any(-1) | (pos > upper[:, None]).any(-1)
    want_h = torch.as_tensor(g[f'dm_{tag}_health0'].copy()).cuda()
    want_h += -.05*(m.sum(1).float() + outside) - .001
    assert torch.equal(health, want_h) and not reset.any() and torch.equal(dead, health <= 0)
    # the revive half: everybody flagged starts from (1, 0) whatever they held
    health2, damage2 = torch.full_like(health, -3.), torch.full_like(damage, 7.)
    flagged = torch.rand((F, A), device='cuda') < .5
    dead2 = flagged.clone()
    reset2, reward2, obs2 = cuda.deathmatch_shoot(centre, positions, upper, health2, damage2, dead2)
    base_h = torch.where(flagged, torch.ones_like(health), torch.full_like(health, -3.))
    base_d = torch.where(flagged, torch.zeros_like(damage), torch.full_like(damage, 7.))
    assert torch.equal(reset2, flagged) and torch.equal(reward2, reward)
    assert torch.equal(health2, base_h + (-.05*(m.sum(1).float() + outside) - .001)) and torch.equal(damage2, base_d + .05*m.sum(2).float())
    assert torch.equal(dead2, health2 <= 0) and torch.equal(obs2, health2)


def test_deathmatch_steps_the_same_with_and_without_the_logic_kernel():
    """Two Deathmatch envs from the same seeds, one settling each frame with the tensor ops, one with the single launch: the same
    observations, rewards, resets and agent state step after step - through deaths (health pushed down by hand so that agents
    die and respawn inside the physics launch) - and the same matchings when asked."""
    from megastep_amd.demo import Deathmatch
    from megastep_amd import arrdict, cubicasa
    gs = cubicasa.sample(6, n_unique=16)

    def rollout(fused):
        torch.manual_seed(3); np.random.seed(3)
        env = Deathmatch(24, 4, geometries=gs, fused=fused)
        assert env._fused == fused
        torch.manual_seed(4)
        frames = [env.reset()]
        for t in range(25):
            if t in (5, 11, 12):
                env._health[t % 6, (t + 1) % 4] = -.5          # somebody dies: back at full health after the next step's respawn
                if fused:
                    env._dead[t % 6, (t + 1) % 4] = True       # (the kernel's list is what the fused step reads)
            acts = torch.randint(0, 7, (24, 1), device='cuda', generator=torch.Generator('cuda').manual_seed(100 + t))
            frames.append(env.step(arrdict.arrdict(actions=acts)))
            frames[-1]['matchings'] = env.matchings.clone()
            frames[-1]['state'] = torch.cat([env.core.agents.positions.reshape(-1), env._health.reshape(-1), env._damage.reshape(-1)])
        return frames

    a, b = rollout(False), rollout(True)
    resets = 0
    for t, (fa, fb) in enumerate(zip(a, b)):
        assert torch.equal(fa.reset, fb.reset) and torch.equal(fa.reward, fb.reward), t
        for k in ('rgb', 'd', 'imu', 'health'):
            torch.testing.assert_close(fa.obs[k], fb.obs[k], rtol=0, atol=1e-6, msg=f'{k} at step {t}')
        if t:
            assert torch.equal(fa.matchings, fb.matchings)
            torch.testing.assert_close(fa.state, fb.state, rtol=0, atol=1e-6)
            resets += int(fa.reset.sum())
    assert a[0].reset.all() and resets >= 3


def test_envs_soak():
    """A few hundred steps of each demo env under random actions (the one-launch game logic, respawns inside the physics launch, spawn
    choices drawn ahead): observations finite and in range throughout, agents inside their floorplans' extents, Deathmatch's health
    never above 1 and its dead back at full health a step later, Explorer's potential never decreasing within an episode, resets
    and rewards happening."""
    from megastep_amd.demo import Deathmatch, Explorer, Minimal
    from megastep_amd import arrdict, cubicasa
    torch.manual_seed(7); np.random.seed(7)
    gs = cubicasa.sample(12, n_unique=16)
    dm = Deathmatch(48, 4, geometries=gs)
    w = dm.reset()
    hi = dm._bounds.max() + 2
    resets = rewards = 0
    for t in range(400):
        if t % 50 == 10:
            dm._health[t % 12, t % 4] = -1.; dm._dead[t % 12, t % 4] = True        # (somebody is killed: hits alone take hundreds of steps)
        w = dm.step(_decision(dm, 48))
        assert torch.isfinite(w.obs.rgb).all() and torch.isfinite(w.obs.d).all() and torch.isfinite(w.obs.imu).all()
        assert (w.obs.d >= 0).all() and (w.obs.d <= 1).all() and (w.obs.rgb >= 0).all() and (w.obs.rgb <= 1.0001).all()
        assert (dm._health <= 1).all() and (w.reward >= 0).all()
        pos = dm.core.agents.positions
        assert (pos > -2).all() and (pos < hi).all()
        if w.reset.any():
            assert (dm._health.reshape(-1)[w.reset] > .7).all()          # revived this step: full health less at most this frame's wounds
        resets += int(w.reset.sum()); rewards += float(w.reward.sum())
    assert resets >= 8 and rewards > 0
    ex = Explorer(24, geometries=cubicasa.sample(24, n_unique=32))
    w = ex.reset()
    last = ex._potential.clone()
    resets = 0
    for t in range(300):
        if t % 60 == 20:
            ex._lengths[t % 24] = 10_000                               # an episode is ended (200 steps + a step per texel otherwise)
        w = ex.step(_decision(ex, 24))
        for k in w.obs:
            assert torch.isfinite(w.obs[k]).all(), k
        assert (w.reward >= 0).all() and (w.reward[w.reset] == 0).all()
        grown = ex._potential >= torch.where(w.reset, torch.zeros_like(last), last)
        assert grown.all()
        last = ex._potential.clone()
        resets += int(w.reset.sum())
    assert resets >= 4 and last.min() > 0
    mi = Minimal(32)
    mi.reset()
    for t in range(200):
        w = mi.step(_decision(mi, 32))
        assert torch.isfinite(w.obs).all()
    assert (mi.core.agents.positions > 1).all() and (mi.core.agents.positions < 6).all()


def test_observation_modules_against_a_torch_restatement():
    """Depth/RGB on a real render equal their definition (modules.py:170-184,211-224) applied to the raw outputs."""
    from megastep_amd import core, cubicasa, modules, scene, cuda
    np.random.seed(0)
    gs = cubicasa.sample(4, n_unique=16)
    c = core.Core(scene.scenery(gs, 2, random=np.random.RandomState(0)), res=64, fov=100)
    modules.RandomSpawns(gs, c)(c.agent_full(True))
    r = modules.render(c)
    assert r.screen.shape == (4, 2, 3, 1, 64) and r.indices.shape == (4, 2, 1, 64)
    d = modules.Depth(c, subsample=4, max_depth=10)(r)
    want = 1 - ((r.distances - c.agent_radius)/10).clamp(0, 1)
    torch.testing.assert_close(d, want.view(4, 2, 1, 16, 4).mean(-1).unsqueeze(3))
    rgb = modules.RGB(c, subsample=4)(r)
    torch.testing.assert_close(rgb, r.screen.view(4, 2, 3, 1, 16, 4).mean(-1))
    raw = cuda.render(c.scenery, c.agents)
    torch.testing.assert_close(raw.screen.permute(0, 1, 3, 2).unsqueeze(3), r.screen)


def test_explorer_reward_equals_the_reference_formula():
    """The incremental bookkeeping must give the reference's numbers: potential = seen texels per env
    (explorer.py:45-58), reward = its increase / 64, respawned envs forget."""
    from megastep_amd.demo import Explorer
    from megastep_amd import cubicasa
    torch.manual_seed(1); np.random.seed(1)
    env = Explorer(8, geometries=cubicasa.sample(8, n_unique=16), fused=False)      # (the tensor-op books; the one-launch ones: next test)
    env.reset()
    prev = env._potential.clone()
    for step in range(30):
        if step == 12:
            env._lengths[3] = 10_000                          # force a respawn of env 3
        world = env.step(_decision(env, 8))
        potential = torch.zeros(8, device='cuda').scatter_add_(0, env._tex_to_env, env._seen.float())
        torch.testing.assert_close(env._potential, potential)
        want = (potential - torch.where(world.reset, torch.zeros_like(prev), prev))/64
        want[world.reset] = 0.
        if step != 12:
            torch.testing.assert_close(world.reward, want)
        prev = potential
    assert world.reset.sum() == 0 and env._potential.min() > 0


@pytest.mark.parametrize('depth_only', [False, True])
def test_explorer_steps_the_same_with_and_without_the_books_kernel(depth_only):
    """Two Explorers from the same seeds, one keeping its books between frames with the tensor ops, one with the single launch
    (ms_explorer_books): the same observations, rewards, resets, potentials and seen-masks step after step, through respawns -
    two forced (an episode length pushed past its limit: the tensor ops apply the rule at the top of the step, the kernel at
    the end of the step before, so the push comes a step earlier there) and the reference's formula for the rest."""
    from megastep_amd.demo import Explorer
    from megastep_amd import arrdict, cubicasa
    gs = cubicasa.sample(8, n_unique=16)

    def rollout(fused):
        torch.manual_seed(5); np.random.seed(5)
        env = Explorer(8, geometries=gs, fused=fused, depth_only=depth_only)
        assert env._fused == fused
        torch.manual_seed(6)
        frames = [env.reset()]
        for t in range(24):
            for when, who in ((7, 2), (15, 5)):
                if t == (when - 1 if fused else when):
                    env._lengths[who] = 10_000
            acts = torch.randint(0, 7, (8, 1), device='cuda', generator=torch.Generator('cuda').manual_seed(200 + t))
            w = env.step(arrdict.arrdict(actions=acts))
            w['potential'], w['seen'] = env._potential.clone(), env._seen.clone()
            w['pose'] = torch.cat([env.core.agents.positions.reshape(-1), env.core.agents.angles.reshape(-1)])
            frames.append(w)
        return frames

    a, b = rollout(False), rollout(True)
    for t, (fa, fb) in enumerate(zip(a, b)):
        assert torch.equal(fa.reset, fb.reset), t
        torch.testing.assert_close(fa.reward, fb.reward, rtol=0, atol=0, msg=f'reward at step {t}')
        for k in fa.obs:
            torch.testing.assert_close(fa.obs[k], fb.obs[k], rtol=0, atol=1e-6, msg=f'{k} at step {t}')
        if t:
            assert torch.equal(fa.potential, fb.potential) and torch.equal(fa.seen, fb.seen) and torch.equal(fa.pose, fb.pose), t
    assert a[0].reset.all() and sum(int(f.reset.sum()) for f in a[1:]) == 2 and a[8].reset[2] and a[16].reset[5]
    assert (torch.stack([f.reward for f in a[1:]]) > 0).any()


@pytest.mark.parametrize('cls,kwargs', [('MomentumMovement', dict(accel=5, ang_accel=180, decay=.125)),
                                        ('SimpleMovement', dict(speed=10, ang_speed=180))])
def test_movement_inside_the_physics_launch_equals_the_tensor_ops(cls, kwargs):
    """the movement modules' velocity update, done by the physics kernel's prologue, against the same
    update done with the reference's tensor ops (modules.py:57-66,106-118) followed by a plain physics call."""
    from megastep_amd import core, cubicasa, cuda, modules, scene
    np.random.seed(2); torch.manual_seed(2)
    gs = cubicasa.sample(16, n_unique=16)
    c = core.Core(scene.scenery(gs, 3, random=np.random.RandomState(0)), res=64, fov=100)
    modules.RandomSpawns(gs, c)(c.agent_full(True))
    mover = getattr(modules, cls)(c, **kwargs)
    keep = 1 - kwargs['decay'] if 'decay' in kwargs else 0.
    for step in range(6):
        actions = torch.randint(0, 7, (16, 3), device='cuda')
        if step == 3:
            c.agents.velocity[0, 0] = float('inf')                      # keep = 0 must assign, not blend
        # the reference's way, on a copy of the state
        ref = cuda.Agents(*(t.clone() for t in (c.agents.angles, c.agents.positions, c.agents.angvelocity, c.agents.velocity)))
        delta = mover._actionset[actions]
        if keep == 0:
            ref.angvelocity[:] = delta.angvelocity
            ref.velocity[:] = modules.to_global_frame(ref.angles, delta.velocity)
        else:
            ref.angvelocity[:] = keep*ref.angvelocity + delta.angvelocity
            ref.velocity[:] = keep*ref.velocity + modules.to_global_frame(ref.angles, delta.velocity)
        moved_v, moved_w = ref.velocity.clone(), ref.angvelocity.clone()
        p_ref = cuda.physics(c.scenery, ref)
        # ours: one launch
        from megastep_amd import arrdict
        p = mover(arrdict.arrdict(actions=actions))
        for name in ('angles', 'positions', 'angvelocity', 'velocity'):
            a, b = getattr(c.agents, name), getattr(ref, name)
            both = torch.isfinite(a) & torch.isfinite(b)
            assert torch.equal(torch.isfinite(a), torch.isfinite(b)), name
            torch.testing.assert_close(a[both], b[both], rtol=0, atol=2e-6, msg=name)
        both = torch.isfinite(p.progress) & torch.isfinite(p_ref.progress)
        torch.testing.assert_close(p.progress[both], p_ref.progress[both], rtol=0, atol=2e-6)
        assert (p.progress < 1).any() or step < 2
        if keep != 0:
            c.agents.velocity[0, 0] = 0.


@pytest.mark.parametrize('after,n_agents', [(False, 3), (True, 3), (False, 1), (True, 70), (False, 66)])
def test_respawn_and_imu_inside_the_physics_launch_equal_the_tensor_ops(after, n_agents):
    """respawn (before the step - Deathmatch's order - or after it - Explorer's) and the IMU reading, done
    by the physics launch, against the modules' tensor ops (modules.py:263-270,312-326) around a plain movement call.
    More than 64 agents per env: the ones past a wavefront's lanes go through memory."""
    from megastep_amd import arrdict, core, cubicasa, cuda, modules, scene
    np.random.seed(3); torch.manual_seed(3)
    A = n_agents
    gs = cubicasa.sample(24, n_unique=32)
    c = core.Core(scene.scenery(gs, A, random=np.random.RandomState(0)), res=32, fov=100)
    spawner = modules.RandomSpawns(gs, c)
    spawner(c.agent_full(True))
    mover, imu = modules.MomentumMovement(c), modules.IMU(c)
    for step in range(8):
        actions = torch.randint(0, 7, (24, A), device='cuda')
        reset = torch.rand((24, A), device='cuda') < (.3 if step % 2 else 0.)
        request = spawner.draw(reset, after=after)
        # the reference's way, on a copy of the state: tensor ops around the fused movement + physics
        ref = cuda.Agents(*(t.clone() for t in (c.agents.angles, c.agents.positions, c.agents.angvelocity, c.agents.velocity)))
        if not after:
            modules._respawn(ref, request)
        table = torch.cat([mover._actionset.velocity, mover._actionset.angvelocity[:, None]], 1).contiguous()
        p_ref = cuda.physics(c.scenery, ref, movement=(actions, table, 1 - mover.decay))
        if after:
            modules._respawn(ref, request)
        imu_ref = torch.cat([ref.angvelocity[..., None]/imu.ang_scale,
                             modules.to_local_frame(ref.angles, ref.velocity)/imu.speed_scale], -1)
        # ours: one launch
        p = mover(arrdict.arrdict(actions=actions), respawn=request, imu=imu)
        reading = imu()
        assert imu._pending is None and reading.shape == (24, A, 3)
        for name in ('angles', 'positions', 'angvelocity', 'velocity'):
            torch.testing.assert_close(getattr(c.agents, name), getattr(ref, name), rtol=0, atol=2e-6, msg=name)
        torch.testing.assert_close(p.progress, p_ref.progress, rtol=0, atol=2e-6)
        torch.testing.assert_close(reading, imu_ref, rtol=0, atol=2e-6)
        if reset.any():
            assert (c.agents.velocity[reset] == 0).all() or not after
        # the renderer must see the respawned pose (the heading cache follows the new angle)
        r = cuda.render(c.scenery, c.agents)
        r_ref = cuda.render(c.scenery, ref)
        assert torch.equal(r.indices, r_ref.indices)
    torch.testing.assert_close(imu(), torch.cat([c.agents.angvelocity[..., None]/imu.ang_scale,
                               modules.to_local_frame(c.agents.angles, c.agents.velocity)/imu.speed_scale], -1))


def test_lifespans_inside_the_physics_launch_equal_the_module():
    """modules.py:361-366 inside the launch: ages tick, the expired join the respawn mask and get a fresh maximum."""
    from megastep_amd import core, cubicasa, cuda, modules, scene
    np.random.seed(4); torch.manual_seed(4)
    gs = cubicasa.sample(16, n_unique=16)
    c = core.Core(scene.scenery(gs, 2, random=np.random.RandomState(0)), res=32)
    spawner = modules.RandomSpawns(gs, c)
    spawner(c.agent_full(True))
    life = modules.RandomLifespans(c, max_lifespan=8)
    ages, maxima = life._lifespans.clone(), life._max_lifespans.clone()
    respawned = 0
    for step in range(30):
        outside = torch.rand((16, 2), device='cuda') < .05
        fresh = torch.randint(life.min_lifespan, life.max_lifespan, (16, 2), device='cuda', dtype=torch.int32)
        # the module's rules, by hand
        want_ages = ages + 1
        want_reset = (want_ages >= maxima) | outside
        want_ages = torch.where(want_reset, torch.zeros_like(want_ages), want_ages)
        want_max = torch.where(want_reset, fresh, maxima)
        request = spawner.draw(outside.clone())
        before = c.agents.positions.clone()
        cuda.physics(c.scenery, c.agents, respawn=request, lifespans=dict(lifespans=ages, max_lifespans=maxima, fresh=fresh))
        assert torch.equal(request['mask'], want_reset) and torch.equal(ages, want_ages) and torch.equal(maxima, want_max)
        moved = (c.agents.positions != before).any(-1)
        assert torch.equal(moved & want_reset, moved)                       # velocities are zero: only respawns move anyone
        respawned += int(want_reset.sum())
    assert respawned > 40


def test_crosshair_ids_from_the_render_kernel():
    """deathmatch.py:54-58,74-80: the two centre pixels' opponent ids written by the render kernel equal what the
    full-resolution hit lines give; and a Deathmatch stepping on them runs."""
    from megastep_amd import core, cubicasa, modules, scene
    from megastep_amd.demo.envs import deathmatch
    np.random.seed(5); torch.manual_seed(5)
    gs = cubicasa.sample(32, n_unique=32)
    c = core.Core(scene.scenery(gs, 4, random=np.random.RandomState(0)), res=128, fov=70)
    modules.RandomSpawns(gs, c)(c.agent_full(True))
    # put agents in front of each other in some envs so that crosshairs are not empty
    for e in range(0, 32, 2):
        c.agents.positions[e, 1] = c.agents.positions[e, 0] + torch.tensor([.5, 0.], device='cuda')
        c.agents.angles[e, 0], c.agents.angles[e, 1] = 0., 180.
    rgb, depth = modules.RGB(c, n_agents=1, subsample=4), modules.Depth(c, n_agents=1, subsample=4)
    frame = modules.render(c, observers=(rgb, depth), fields=('indices',), centre=True)
    want = deathmatch.crosshair_matrix(frame.indices, 8, 4, 4)
    got = (frame.centre[..., None] == torch.arange(4, device='cuda')).any(-2)
    assert torch.equal(got, want) and want.any()
    assert frame.centre.shape == (32, 4, 2) and frame.centre.min() >= -1 and frame.centre.max() < 4
    lean = modules.render(c, observers=(rgb, depth), fields=(), centre=True)
    assert torch.equal(lean.centre, frame.centre) and 'indices' not in lean


def test_first_sight_bookkeeping_in_the_render_kernel():
    """explorer.py:34-58: after every frame the stamped texels are exactly the texels under a ray (so far), and the
    tally is their number, per env; forgetting an env starts it over."""
    from megastep_amd import core, cubicasa, cuda, modules, scene
    from megastep_amd.demo.envs import explorer
    np.random.seed(6); torch.manual_seed(6)
    gs = cubicasa.sample(16, n_unique=16)
    c = core.Core(scene.scenery(gs, 1, random=np.random.RandomState(0)), res=256, fov=130)
    spawner = modules.RandomSpawns(gs, c)
    spawner(c.agent_full(True))
    memory = explorer.SeenTexels(c.scenery, 16)
    seen = torch.zeros(c.scenery.textures.vals.shape[0], dtype=torch.bool, device='cuda')
    rng = np.random.RandomState(0)
    for step in range(12):
        if step == 6:
            which = torch.zeros(16, dtype=torch.bool, device='cuda'); which[[2, 9]] = True
            memory.forget(which)
            seen[which[memory.texel_env]] = False
        from tests import util
        util.random_velocities(c, rng)
        cuda.physics(c.scenery, c.agents)
        frame = modules.render(c, fields=('indices', 'locations'), seen=memory.books)
        texels = explorer.texels_hit(c.scenery, frame)
        seen[texels.flatten()] = True               # (as the reference does it: a miss, -1, marks the LAST texel - explorer.py:36,47)
        want = torch.zeros(16, device='cuda').scatter_add_(0, memory.texel_env, seen.float())
        assert torch.equal(memory.mask(), seen)
        torch.testing.assert_close(memory.count, want)
    assert memory.count.min() > 20
    gained = memory.gained()
    assert (gained >= 0).all() and memory.gained().sum() == 0


def test_env_steps_replayed_as_a_hip_graph():
    """graphs.GraphedStep: the captured step takes the actions it is given, moves the world on and keeps producing sane
    observations; an env whose agents all hold still under the no-op action does so under the graph too."""
    from megastep_amd import arrdict, cubicasa, graphs
    from megastep_amd.demo import Explorer
    geometries = cubicasa.sample(16, n_unique=16)
    env = graphs.GraphedStep(Explorer(16, geometries=geometries))
    env.reset()
    before = env.core.agents.positions.clone()
    forward = arrdict.arrdict(actions=torch.ones((16, 1), dtype=torch.long, device='cuda'))
    for _ in range(4):
        world = env.step(forward)
    moved = (env.core.agents.positions - before).norm(dim=-1)
    assert (moved > .05).float().mean() > .5 and torch.isfinite(world.obs.rgb).all() and world.obs.rgb.shape == (16, 1, 3, 1, 64)
    noop = arrdict.arrdict(actions=torch.zeros((16, 1), dtype=torch.long, device='cuda'))
    for _ in range(60):
        env.step(noop)                                                     # momentum decays away
    at_rest = env.core.agents.positions.clone()
    env.step(noop)
    assert (env.core.agents.positions - at_rest).norm(dim=-1).max() < 2e-3


def test_fused_envs_replayed_as_hip_graphs():
    """graphs.GraphedStep around the envs with the one-launch game logic: Deathmatch (the respawn mask written by one step's logic
    kernel and read by the next step's physics launch, both inside the captured step; the spawn draw stays torch's graph-safe
    randint under capture) and Minimal (its whole step one launch): the replayed steps keep the books the eager envs keep."""
    from megastep_amd import arrdict, cubicasa, graphs
    from megastep_amd.demo import Deathmatch, Minimal
    torch.manual_seed(11); np.random.seed(11)
    env = graphs.GraphedStep(Deathmatch(32, 4, geometries=cubicasa.sample(8, n_unique=16)))
    env.reset()
    forced = 0
    for t in range(120):
        if t in (20, 60):
            env.env._health[t % 8, 1] = -1.; env.env._dead[t % 8, 1] = True      # killed: the next replay respawns it at full health
        w = env.step(_decision(env.env, 32))
        assert torch.isfinite(w.obs.rgb).all() and torch.isfinite(w.obs.health).all() and (env.env._health <= 1).all()
        if t in (20, 60):
            forced += int(w.reset.reshape(8, 4)[t % 8, 1])
            assert env.env._health[t % 8, 1] > .7
    assert forced == 2
    assert (env.env._health < 1 - 50*.001).any()                             # the tick damage of the replayed steps adds up
    mini = graphs.GraphedStep(Minimal(16))
    mini.reset()
    start = mini.core.agents.positions.clone()
    forward = arrdict.arrdict(actions=torch.ones((16, 1), dtype=torch.long, device='cuda'))
    for _ in range(6):
        w = mini.step(forward)
    assert torch.isfinite(w.obs).all() and ((mini.core.agents.positions - start).norm(dim=-1) > .2).all()


def test_rays_that_miss_mark_the_last_texel_like_the_reference():
    """explorer.py:36,47: a missed ray's texel index is -1 and `_seen[-1] = True` marks the scenery's LAST texel, to the
    credit of the last env. Open worlds - a lone wall per env, most rays see nothing - with and without colour, groups of
    rays per wave pinned to 1 and 4: the books the kernel keeps equal the reference's formula on the full planes."""
    from megastep_amd import _lib, cuda, modules
    from megastep_amd.demo.envs import explorer
    from tests.test_gpu_parity import _custom_world
    walls = [np.array([[[3., 1.], [3., 2.]]]), np.array([[[2., 3.], [4., 3.]], [[4., 3.], [4., 3.5]]]), np.array([[[1., 1.], [1., 1.4]]])]
    c = _custom_world(walls, 1, 256, 130, [[[2., 1.5]], [[3., 2.]], [[2., 2.]]], [[0.], [90.], [0.]])
    T = c.scenery.textures.vals.shape[0]
    depth, rgb = modules.Depth(c, subsample=4), modules.RGB(c, subsample=4)
    for groups, observers in ((1, (rgb, depth)), (4, (depth,)), (0, (rgb, depth))):
        _lib.lib().ms_debug_ray_groups(groups)
        try:
            memory = explorer.SeenTexels(c.scenery, 3)
            seen = torch.zeros(T, dtype=torch.bool, device='cuda')
            for step, angles in enumerate(([0., 90., 0.], [30., 60., 180.], [-40., 120., 90.])):
                c.agents.angles[:] = torch.as_tensor(angles, device='cuda')[:, None]
                if step == 2:
                    which = torch.tensor([False, False, True], device='cuda')       # the last env forgets - its last texel too
                    memory.forget(which)
                    seen[which[memory.texel_env]] = False
                full = modules.render(c, fields=('indices', 'locations'))
                modules.render(c, observers=observers, fields=(), seen=memory.books)
                texels = explorer.texels_hit(c.scenery, full)
                assert (texels < 0).any() and (texels >= 0).any()
                seen[texels.flatten()] = True
                want = torch.zeros(3, device='cuda').scatter_add_(0, memory.texel_env, seen.float())
                assert torch.equal(memory.mask(), seen) and bool(seen[-1])
                torch.testing.assert_close(memory.count, want)
        finally:
            _lib.lib().ms_debug_ray_groups(0)
