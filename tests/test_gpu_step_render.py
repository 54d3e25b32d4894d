"""ms_step_render / cuda.step_render: a step of the hot path (ms_physics then ms_render, the reference's wrappers.cpp:69 + :82)
as one call - ONE LAUNCH for single-agent worlds of up to 64 rays (BASELINE config 2), where an agent is a single wavefront
that runs its env's physics and renders from the pose it ends on - against the two calls, bit for bit, and against the oracle."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _world(n_envs, n_agents, res, fov, seed=0, toy=None, large=False, oblique=False, grid=True):
    from megastep_amd import core, cubicasa, cuda, scene, toys
    np.random.seed(seed)
    geometries = n_envs*[getattr(toys, toy)()] if toy else cubicasa.sample(n_envs, n_unique=32, seed=seed + 1, large=large, oblique=oblique)
    scenery = scene.scenery(geometries, n_agents, device='cuda', random=np.random.RandomState(seed), bake=False)
    cuda.bake(scenery, wall_grid=grid)
    c = core.Core(scenery, res=res, fov=fov, fps=10)
    util.spawn(c, geometries, seed=seed)
    return c, geometries


def _state(c):
    a = c.agents
    return [t.clone() for t in (a.angles, a.positions, a.angvelocity, a.velocity, c.scenery.lines.vals)]


def _restore(c, state):
    a = c.agents
    for t, s in zip((a.angles, a.positions, a.angvelocity, a.velocity, c.scenery.lines.vals), state):
        t.copy_(s)


def _same(x, y):
    return torch.equal(torch.nan_to_num(x.float(), nan=-7.), torch.nan_to_num(y.float(), nan=-7.))


#: (an A/B build asked for one of its older raycasts - `make ab`, MEGASTEP_RENDER_IMPL=pairs|seq - has no one-launch step: the two
#: launches then, with the same results, which is all these tests compare)
OLDER_RAYCAST = bool(__import__('os').environ.get('MEGASTEP_RENDER_IMPL'))


def _fused():
    from megastep_amd import _lib
    return bool(_lib.lib().ms_debug_last_step_fused()) or OLDER_RAYCAST


def _both_ways(c, steps, rng, fields=None, pooled=None, speed=(4., 40.), expect_fused=True, prepare=None):
    """`steps` steps from the same start, once as cuda.physics + cuda.render and once as cuda.step_render: every output and the
    whole agent state equal after every step."""
    from megastep_amd import cuda
    start = _state(c)
    vels = []
    for i in range(steps):
        util.random_velocities(c, rng, speed=speed[i % len(speed)])
        if prepare:
            prepare(c, i)
        vels.append((c.agents.velocity.clone(), c.agents.angvelocity.clone()))
    frames = []
    for fused in (False, True):
        _restore(c, start)
        out = []
        for i, (v, w) in enumerate(vels):
            c.agents.velocity.copy_(v); c.agents.angvelocity.copy_(w)
            if prepare:
                prepare(c, i)
            if fused:
                p, r = cuda.step_render(c.scenery, c.agents, fields=fields, pooled=pooled)
                assert _fused() == (expect_fused or OLDER_RAYCAST)
            else:
                p = cuda.physics(c.scenery, c.agents)
                r = cuda.render(c.scenery, c.agents, fields=fields, pooled=pooled)
            out.append((p.progress.clone(), r, _state(c), c.agents._headings.clone()))
        frames.append(out)
    for i, ((pa, ra, sa, ha), (pb, rb, sb, hb)) in enumerate(zip(*frames)):
        assert _same(pa, pb), f'progress differs at step {i}'
        for k, (x, y) in enumerate(zip(sa, sb)):
            assert _same(x, y), f'state tensor {k} differs at step {i}'
        assert _same(ha, hb), f'heading cache differs at step {i}'
        for f in ('indices', 'locations', 'dots', 'distances', 'screen', 'obs_rgb', 'obs_depth', 'obs_centre'):
            x, y = getattr(ra, f), getattr(rb, f)
            assert (x is None) == (y is None), f
            if x is not None:
                assert _same(x, y), f'{f} differs at step {i}'
    return frames[1]


@pytest.mark.parametrize('n_envs,res,fov,kw', [
    (64, 64, 130., {}),
    (24, 33, 90., {}),                                   # a ragged group of rays
    (16, 8, 160., dict(toy='column')),
    (12, 64, 130., dict(large=True)),
    (24, 64, 130., dict(oblique=True)),
    (16, 64, 130., dict(grid=False)),                    # no wall grid: both halves meet every wall
    (8, 64, 170., {}),                                   # a view wider than the vis lists allow: the grid serves physics only - two launches
])
def test_one_launch_equals_the_two_calls(n_envs, res, fov, kw):
    c, _ = _world(n_envs, 1, res, fov, seed=3, **kw)
    rng = np.random.RandomState(5)
    one_launch = fov <= 165.
    frames = _both_ways(c, 6, rng, expect_fused=one_launch)
    progress = torch.stack([f[0] for f in frames])
    assert (progress < 1).any() and (progress == 1).any(), 'some agents ran into something, some did not'


def test_one_launch_with_every_family_of_outputs():
    """Depth-only (render_kernel<2,1,1,0,1,1>), pooled observations (<2,1,1,1,1,1>) and all five planes (<2,1,0,1,1,1>)."""
    c, _ = _world(32, 1, 64, 130., seed=4)
    rng = np.random.RandomState(6)
    _both_ways(c, 4, rng, fields=('distances',))
    _both_ways(c, 4, rng, fields=('indices', 'dots'))
    _both_ways(c, 4, rng, fields=(), pooled=dict(subsample=4, max_depth=10.))
    _both_ways(c, 4, rng, fields=('screen',), pooled=dict(subsample=2, max_depth=3., rgb=False))


def test_agents_the_near_lists_do_not_cover():
    """Outside the grid, faster than the lists reach, crawling, standing still, at NaN: the wave meets every wall of the env, as
    physics_kernel does for such an env - and renders what ms_render renders from wherever that leaves the agent."""
    c, _ = _world(16, 1, 64, 130., seed=5)

    def prepare(c, i):
        v = c.agents.velocity
        v[0, 0] = torch.tensor([300., 0.], device='cuda')            # 30 m a step
        v[1, 0] = torch.tensor([3e-6, 0.], device='cuda')            # crawling: project()'s 1e-6 reaches walls metres away
        v[2, 0] = 0.
        if i == 0:
            c.agents.positions[3, 0] = torch.tensor([-50., -50.], device='cuda')     # outside the grid
            c.agents.positions[4, 0] = torch.tensor([float('nan'), 2.], device='cuda')
    _both_ways(c, 4, np.random.RandomState(7), prepare=prepare)


def test_other_shapes_take_the_two_launches():
    """Several agents per env, or more than 64 rays: ms_step_render is ms_physics followed by ms_render."""
    for n_agents, res in ((2, 64), (1, 128), (4, 512)):
        c, _ = _world(6, n_agents, res, 70., seed=6)
        _both_ways(c, 3, np.random.RandomState(8), expect_fused=False)


def test_without_the_heading_cache_and_with_out():
    from megastep_amd import cuda
    c, _ = _world(16, 1, 64, 130., seed=7)
    c.agents._use_cache = False
    _both_ways(c, 3, np.random.RandomState(9))
    c.agents._use_cache = True
    rng = np.random.RandomState(10)
    util.random_velocities(c, rng)
    first = cuda.step_render(c.scenery, c.agents)
    want = first[1].distances.clone()
    util.random_velocities(c, rng)
    again = cuda.step_render(c.scenery, c.agents, out=first)
    assert again[0] is first[0] and again[1] is first[1] and not torch.equal(again[1].distances, want)


def test_one_launch_matches_the_oracle_at_c2s_size():
    """BASELINE config 2's shape (4096 x 1 x 64, one floorplan per env) stepped as one launch per step, against the oracle."""
    from megastep_amd import cuda
    import bench
    bench.PLAN_CONTEXT = 'subprocess'
    c, _ = bench.build_world(4096, 1, 64, 130., torch.device('cuda'), seed=1, n_unique=512)
    ref = util.OracleWorld(c)
    ref.pull_baked(c)
    rng = np.random.RandomState(11)
    for step in range(3):
        util.random_velocities(c, rng, speed=6.)
        ref.pull_agents(c)
        p, r = cuda.step_render(c.scenery, c.agents)
        assert _fused()
        prog_ref, agents_ref = ref.physics()
        util.assert_physics_matches(c, p, prog_ref, agents_ref)
        util.assert_render_matches(c, r, ref.render())


def test_steps_replayed_as_a_hip_graph():
    from megastep_amd import cuda
    c, _ = _world(64, 1, 64, 130., seed=8)
    rng = np.random.RandomState(12)
    util.random_velocities(c, rng, speed=3.)
    start = _state(c)
    out = cuda.step_render(c.scenery, c.agents)
    eager = []
    _restore(c, start)
    for _ in range(5):
        cuda.step_render(c.scenery, c.agents, out=out)
        eager.append((out[1].distances.clone(), c.agents.positions.clone()))
    _restore(c, start)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cuda.step_render(c.scenery, c.agents, out=out)
    _restore(c, start)
    for d, p in eager:
        g.replay()
        assert torch.equal(out[1].distances, d) and torch.equal(c.agents.positions, p)


@pytest.mark.parametrize('keep,after', [(0., False), (.875, False), (.875, True)])
def test_one_launch_with_the_movement_prologue_and_the_envs_bookkeeping(keep, after):
    """ms_move_step_render: what ms_step_physics runs around the step - the movement modules' velocity update, lifespans, masked
    respawns (before the step or after it), the IMU reading - inside the one launch too: agent state, progress, the IMU
    observation, the lifespans' books and the respawn mask they report into, and every render output equal, bit for bit, to
    cuda.physics(movement=, respawn=, lifespans=, imu=) followed by cuda.render, eight steps."""
    from megastep_amd import cuda, modules
    c, geometries = _world(48, 1, 64, 130., seed=9)
    N = c.n_envs
    rng = np.random.RandomState(13)
    spawner = modules.RandomSpawns(geometries, c)
    table = modules._table(modules._actionset(c, 5, 180))
    g = torch.Generator('cuda').manual_seed(3)
    acts = torch.randint(0, 7, (8, N, 1), device='cuda', generator=g)
    masks = torch.rand((8, N, 1), device='cuda', generator=g) < .15
    choices = torch.randint(0, 1, (8, N, 1), device='cuda', generator=g)            # (the reference's quirk: spawns.shape[1] = n_agents options)
    fresh = torch.randint(3, 9, (8, N, 1), device='cuda', generator=g, dtype=torch.int32)
    util.random_velocities(c, rng, speed=2.)
    start = _state(c)
    results = []
    for fused in (False, True):
        _restore(c, start)
        ages = torch.zeros((N, 1), dtype=torch.int32, device='cuda')
        maxima = torch.full((N, 1), 5, dtype=torch.int32, device='cuda')
        out = []
        for t in range(8):
            mask = masks[t].clone()
            request = dict(mask=mask, choices=choices[t], positions=spawner._spawns.positions, angles=spawner._spawns.angles, after=after)
            life = dict(lifespans=ages, max_lifespans=maxima, fresh=fresh[t])
            imu = torch.full((N, 1, 3), float('nan'), device='cuda')
            kw = dict(movement=(acts[t], table, keep), respawn=request, lifespans=life, imu=(imu, 360., 10.))
            if fused:
                p, r = cuda.step_render(c.scenery, c.agents, **kw)
                assert _fused()
            else:
                p = cuda.physics(c.scenery, c.agents, **kw)
                r = cuda.render(c.scenery, c.agents)
            out.append((p.progress.clone(), r, _state(c), c.agents._headings.clone(), imu, mask, ages.clone(), maxima.clone()))
        results.append(out)
    respawned = 0
    for t, (a, b) in enumerate(zip(*results)):
        assert _same(a[0], b[0]), f'progress at step {t}'
        for k, (x, y) in enumerate(zip(a[2], b[2])):
            assert _same(x, y), f'state tensor {k} at step {t}'
        assert _same(a[3], b[3]) and _same(a[4], b[4]), f'heading cache / imu at step {t}'
        assert torch.equal(a[5], b[5]) and torch.equal(a[6], b[6]) and torch.equal(a[7], b[7]), f'lifespans / mask at step {t}'
        for f in ('indices', 'locations', 'dots', 'distances', 'screen'):
            assert _same(getattr(a[1], f), getattr(b[1], f)), f'{f} at step {t}'
        respawned += int(a[5].sum())
    assert respawned > 20 and torch.isfinite(results[1][-1][4]).all()


def test_the_tutorial_env_steps_in_one_launch():
    """demo.Minimal (the reference's tutorial env, demo/envs/minimal.py: SimpleMovement, physics, render of 64 rays): its step
    through modules.move_render - one launch - returns what the movement module followed by the render does."""
    from megastep_amd import arrdict
    from megastep_amd.demo import Minimal
    rollouts = []
    for fused in (True, False):
        torch.manual_seed(2); np.random.seed(2)
        env = Minimal(64)
        torch.manual_seed(3)
        frames = [env.reset().obs.clone()]
        for t in range(12):
            decision = arrdict.arrdict(actions=torch.randint(0, 7, (64, 1), device='cuda', generator=torch.Generator('cuda').manual_seed(50 + t)))
            if fused:
                frames.append(env.step(decision).obs.clone())
                assert _fused()
            else:
                env.movement(decision)
                frames.append(env._world().obs.clone())
        frames.append(env.core.agents.positions.clone())
        rollouts.append(frames)
    for t, (x, y) in enumerate(zip(*rollouts)):
        assert _same(x, y), t
    assert (rollouts[0][-1] - 3.).abs().max() > .1
