"""render_kernel's NG - how many 64-ray groups one wave serves (1, 2 or 4; ms_render picks 1 or 4) - and the split of a launch
into wide waves and the one-group waves it ends with change who does the work, never the result: every setting against the oracle and against the others, bit for bit."""
import os

import numpy as np
import pytest
import torch

from tests import util
from tests.test_gpu_parity import _custom_world, _world

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.detach().cpu().numpy().view(np.int32)


@pytest.fixture
def groups():
    from megastep_amd import _lib
    h = _lib.lib()

    def pin(g, tail_envs=0):
        """g groups per wave for all but the last `tail_envs` envs, whose waves serve one (-1: as ms_render sizes that share)"""
        h.ms_debug_ray_groups(g)
        h.ms_debug_ray_group_tail(-1., tail_envs)
    yield pin
    h.ms_debug_ray_groups(0)
    h.ms_debug_ray_group_tail(-1., -1)


def _same(a, b, what):
    for f in ('indices', 'locations', 'dots', 'distances', 'screen'):
        assert np.array_equal(_bits(getattr(a, f)), _bits(getattr(b, f))), (what, f)


@pytest.mark.parametrize('n_agents,res,fov,large', [(4, 64, 130, False), (3, 100, 90, False), (4, 128, 70, False), (2, 256, 130, False),
                                                    (1, 256, 130, True), (4, 512, 70, False), (2, 600, 160, False), (5, 129, 20, False), (1, 1, 90, False)])
def test_every_number_of_ray_groups_per_wave_gives_the_oracles_render(groups, n_agents, res, fov, large):
    from megastep_amd import cuda
    c, _ = _world(3 if large else 6, n_agents, res, fov, seed=31, large=large)
    rng = np.random.RandomState(4)
    ref = util.OracleWorld(c)
    ref.pull_baked(c)
    for step in range(2):
        util.random_velocities(c, rng)
        cuda.physics(c.scenery, c.agents)
        ref.pull_agents(c)
        want = ref.render()
        frames = {}
        for g in (0, 1, 2, 4):                                               # 0: as ms_render picks
            groups(g)
            frames[g] = cuda.render(c.scenery, c.agents)
            util.assert_render_matches(c, frames[g], want)
        for g in (1, 2, 4):
            _same(frames[g], frames[0], (g, step))
        # a launch of wide waves that ends in one-group waves: for its last env, for all but its first, for every env
        for g in (2, 4):
            for tail in (1, c.n_envs - 1, c.n_envs, -1):
                groups(g, tail)
                _same(cuda.render(c.scenery, c.agents), frames[0], (g, step, tail))
        # the pooled observations, the crosshair ids, the colourless instantiation and a partial set of planes too
        if res % 4 == 0 and res >= 8:
            pooled = {}
            for g in (1, 2, 4):
                groups(g)
                p = cuda.render(c.scenery, c.agents, fields=('indices',), pooled=dict(subsample=4, max_depth=8., centre=True))
                d = cuda.render(c.scenery, c.agents, fields=('distances', 'dots'), pooled=dict(subsample=4, max_depth=8., rgb=False))
                pooled[g] = (p.obs_rgb.clone(), p.obs_depth.clone(), p.obs_centre.clone(), p.indices.clone(), d.obs_depth.clone(), d.distances.clone(), d.dots.clone())
                assert np.array_equal(_bits(d.distances), _bits(frames[0].distances)) and np.array_equal(_bits(d.dots), _bits(frames[0].dots))
            for g in (2, 4):
                for x, y in zip(pooled[g], pooled[1]):
                    assert np.array_equal(_bits(x.float()) if x.dtype != torch.float32 else _bits(x), _bits(y.float()) if y.dtype != torch.float32 else _bits(y)), g


def test_stacks_of_coincident_walls_under_every_number_of_groups(groups):
    """The literal folds - over the wave's list in LDS, ray after ray or lane = ray, and over every line of the env when the
    list was worked off more than once - with 128 and 256 rays to a wave: views full of walls within the 1e-4 band of each
    other (kernels.cu:369), in shuffled line order, 512 rays over 40 degrees so that a wave's 256 rays all see the stack; one
    env with 150 such walls, whose (line, ray) pairs - 64 lines x 256 rays a batch - overflow the pair list several times."""
    from megastep_amd import cuda
    rng = np.random.RandomState(5)
    offsets = np.array([0., 0., 2e-5, 5e-5, 9e-5, 1e-4, 1.1e-4, 2e-4, 3e-4, 1e-3])
    envs, pos, ang = [], [], []
    for e in range(24):
        k = 150 if e == 0 else 70 if e == 1 else rng.randint(2, 9)
        xs = 4. + rng.choice(offsets, k)*rng.choice([1, 1, -1], k) + rng.choice([0., 0., .5], k)
        walls = [[[x, 1. + rng.uniform(-.2, .2)], [x, 3. + rng.uniform(-.2, .2)]] for x in xs]
        if e % 3 == 0:
            walls += [[[4., 3.], [2., 3.]], [[2., 3.], [2., 1.]], [[2., 1.], [4., 1.]]]
        if e % 4 == 0:
            walls += [walls[0], [walls[1][1], walls[1][0]]]
        envs.append(np.array(walls)[rng.permutation(len(walls))])
        pos.append([[rng.uniform(2.2, 3.9), rng.uniform(1.5, 2.5)]])
        ang.append([rng.uniform(-20, 20)])
    c = _custom_world(envs, 1, 512, 40, pos, ang)
    ref = util.OracleWorld(c)
    ref.bake(); ref.pull_baked(c); ref.pull_agents(c)
    want = ref.render()
    frames = {}
    for g in (1, 2, 4):
        groups(g)
        frames[g] = cuda.render(c.scenery, c.agents, telemetry=True)
        util.assert_render_matches(c, frames[g], want)
        _, folded_rays, lane_parallel_waves = frames[g]._telemetry[:3].tolist()
        if not os.environ.get('MEGASTEP_RENDER_IMPL'):                  # (the product raycast's counters: an A/B run under an older one has none)
            assert folded_rays > 1000 and lane_parallel_waves > 5, (g, folded_rays, lane_parallel_waves)
    _same(frames[2], frames[1], 2); _same(frames[4], frames[1], 4)


def test_first_sight_books_under_every_number_of_groups(groups):
    from megastep_amd import core, cubicasa, cuda, modules, scene
    from megastep_amd.demo.envs import explorer
    np.random.seed(8); torch.manual_seed(8)
    gs = cubicasa.sample(8, n_unique=16)
    c = core.Core(scene.scenery(gs, 1, random=np.random.RandomState(0)), res=256, fov=130)
    modules.RandomSpawns(gs, c)(c.agent_full(True))
    books = {g: explorer.SeenTexels(c.scenery, 8) for g in (1, 2, 4)}
    rgb, depth = modules.RGB(c, subsample=4), modules.Depth(c, subsample=4)
    rng = np.random.RandomState(1)
    for step in range(5):
        util.random_velocities(c, rng)
        cuda.physics(c.scenery, c.agents)
        obs = {}
        for g in (1, 2, 4):
            groups(g)
            f = modules.render(c, observers=(rgb, depth), fields=(), seen=books[g].books)
            obs[g] = (f.pooled_rgb.clone(), f.pooled_depth.clone())
        for g in (2, 4):
            assert torch.equal(books[g].stamp, books[1].stamp) and torch.equal(books[g].tally, books[1].tally)
            assert torch.equal(obs[g][0], obs[1][0]) and torch.equal(obs[g][1], obs[1][1])
    assert books[4].tally.min() > 20


def test_the_shape_where_ms_render_picks_four_groups_itself(groups):
    """ms_render's own choice (four groups from 256 rays up when there are two and a half rounds of such waves, the last envs
    left to one-group waves) against one group, bit for bit, at a shape on the far side of that rule, with and without colour."""
    from megastep_amd import cuda
    c, _ = _world(4096, 4, 256, 70, seed=8)
    rng = np.random.RandomState(2)
    for _ in range(2):
        util.random_velocities(c, rng)
        cuda.physics(c.scenery, c.agents)
    got = {}
    for g in (0, 1):
        groups(g, -1)
        d = cuda.render(c.scenery, c.agents, fields=('indices', 'locations', 'dots', 'distances'), pooled=dict(subsample=4, max_depth=8., rgb=False, centre=True))
        f = cuda.render(c.scenery, c.agents)
        p = cuda.render(c.scenery, c.agents, fields=(), pooled=dict(subsample=4, max_depth=8.))
        got[g] = [d.indices.clone(), d.locations.clone(), d.dots.clone(), d.distances.clone(), d.obs_depth.float().clone(), d.obs_centre.clone(), f.screen.clone(),
                  f.distances.clone(), p.obs_rgb.float().clone(), p.obs_depth.float().clone()]
    for x, y in zip(got[0], got[1]):
        assert np.array_equal(_bits(x), _bits(y))


@pytest.mark.parametrize('n_envs', [21, 64, 9])
def test_wide_and_single_waves_share_a_launch_whatever_the_envs_per_xcd(groups, n_envs):
    """Every XCD's blocks: wide waves for its envs but the last few, one-group waves for those; XCDs with an env fewer than
    the others let their spare blocks go.  Env counts that split unevenly over the eight, every size of that share."""
    from megastep_amd import cuda
    c, _ = _world(n_envs, 2, 320, 100, seed=12)
    rng = np.random.RandomState(3)
    util.random_velocities(c, rng)
    cuda.physics(c.scenery, c.agents)
    groups(1)
    want = cuda.render(c.scenery, c.agents)
    wantd = cuda.render(c.scenery, c.agents, fields=('distances', 'indices'))
    for g in (2, 4):
        for tail in (0, 1, 8, 9, 16, 17, n_envs, -1):
            groups(g, tail)
            _same(cuda.render(c.scenery, c.agents), want, (g, tail))
            d = cuda.render(c.scenery, c.agents, fields=('distances', 'indices'))
            assert np.array_equal(_bits(d.distances), _bits(wantd.distances)) and np.array_equal(_bits(d.indices), _bits(wantd.indices)), (g, tail)


@pytest.mark.parametrize('n_envs,res,pinned', [(1, 256, 4), (100, 256, 2), (3, 512, 4), (9, 64, 1)])
def test_the_workspace_is_never_written_past_its_declared_size(groups, n_envs, res, pinned):
    """with several ray groups a wave every XCD gets as many blocks as the fullest one needs - more blocks than the
    N A ceil(R/64) entries MS_RENDER_WORKSPACE_INTS provides a queue for (1 env x 4 agents x 256 rays pinned to four groups: 128
    blocks, 16 entries) - and round 4 placed the agents' headings BEHIND the block count: past the end of the caller's buffer.
    The self-contained path (no heading cache: render_prep_kernel fills the workspace) into a workspace of exactly the
    declared size with a fence of sentinels behind it: the fence stands, and the frame is the cached path's bit for bit."""
    from megastep_amd import cuda
    c, _ = _world(n_envs, 4, res, 70, seed=5)
    rng = np.random.RandomState(2)
    util.random_velocities(c, rng)
    cuda.physics(c.scenery, c.agents)
    groups(pinned)
    want = cuda.render(c.scenery, c.agents)                                  # headings from ms_physics' cache: no workspace in play
    got = cuda.render(c.scenery, c.agents, telemetry=True)                   # allocates; we re-point its workspace below
    N, A = c.n_envs, c.n_agents
    words = 18 + N*A*((res + 63)//64) + 2*N*A                                # MS_RENDER_WORKSPACE_INTS(N, A, R)
    fence = 4096
    guarded = torch.full((words + fence,), 0x5a5a5a5a, dtype=torch.int32, device=c.device)
    got._struct.workspace = guarded.data_ptr()
    got = cuda.render(c.scenery, c.agents, telemetry=True, out=got)
    torch.cuda.synchronize()
    assert bool((guarded[words:] == 0x5a5a5a5a).all()), 'ms_render wrote past MS_RENDER_WORKSPACE_INTS words of workspace'
    _same(got, want, (n_envs, res, pinned))
