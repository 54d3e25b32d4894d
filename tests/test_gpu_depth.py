"""The renderer without its shading pass (render_kernel<.,.,1,0>): what a caller gets who asks for no colour - the
reference's Depth reads `distances` only (modules.py:170-184), BASELINE config 2 is Explorer "depth-only" - against the
oracle and against the full renderer's own planes."""
import numpy as np
import pytest
import torch

from tests import util
from tests.test_gpu_parity import _world
from tests.test_gpu_scale import _big_world

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.detach().cpu().numpy().view(np.int32)


def test_depth_only_at_c2_full_size_is_the_oracles_distances_bit_for_bit():
    """4096 envs x 1 agent x 64 rays: `fields=('distances',)` and depth-only pooling, three steps; sampled envs against the
    oracle's raycast - every distance the same binary32, misses +inf - and the pooled depth against modules.Depth on them."""
    from megastep_amd import cuda, modules
    c, _, _ = _big_world(4096, 1, 64, 130, n_distinct=512, fast=True)
    envs = [0, 1, 511, 512, 1023, 2500, 4000, 4095]
    sub = util.OracleSubset(c, envs)
    e = torch.as_tensor(envs, device='cuda')
    depth = modules.Depth(c, subsample=4, max_depth=10.)
    rng = np.random.RandomState(5)
    out = None
    for step in range(3):
        util.random_velocities(c, rng, speed=4. if step % 2 else 30.)
        cuda.physics(c.scenery, c.agents)
        sub.pull_agents(c)
        want = sub.render()
        out = cuda.render(c.scenery, c.agents, fields=('distances',), out=out)
        assert out.indices is None and out.screen is None and out.locations is None and out.dots is None
        np.testing.assert_array_equal(_bits(out.distances[e]), want['distances'].view(np.int32))
        pooled = modules.render(c, observers=(depth,), fields=())
        assert set(pooled.keys()) >= {'pooled_depth'} and 'pooled_rgb' not in pooled and 'screen' not in pooled
        d = torch.as_tensor(want['distances'])
        d_ref = (1 - ((d - c.agent_radius)/10.).clamp(0, 1)).view(len(envs), 1, 16, 4).mean(-1)
        np.testing.assert_allclose(depth(pooled)[e, :, 0, 0].cpu().numpy(), d_ref.numpy(), rtol=0, atol=1e-6)
    assert np.isfinite(want['distances']).mean() > .9


@pytest.mark.parametrize('n_agents,res,fov,grid', [(1, 64, 130, True), (4, 64, 130, True), (3, 100, 90, True), (4, 128, 70, False), (2, 512, 70, True)])
def test_every_colourless_request_leaves_the_full_renderers_bits(monkeypatch, n_agents, res, fov, grid):
    """Any subset of (indices, locations, dots, distances), pooled depth and the crosshair ids, without colour: the planes
    of the full render, bit for bit - also for multi-agent sceneries built WITHOUT a light grid, which the colour path
    can only serve with all five planes and a second launch (there is nothing to light without colour)."""
    from megastep_amd import cuda
    if not grid:
        monkeypatch.setattr(cuda.Scenery, 'LIGHT_GRID', False)
    c, _ = _world(6, n_agents, res, fov, seed=21)
    rng = np.random.RandomState(2)
    util.random_velocities(c, rng)
    cuda.physics(c.scenery, c.agents)
    full = cuda.render(c.scenery, c.agents)
    lines_after = c.scenery.lines.vals.clone()
    ref = util.OracleWorld(c)
    ref.pull_baked(c); ref.pull_agents(c)
    util.assert_render_matches(c, full, ref.render())
    af = c.scenery.n_agents*c.scenery.model.shape[0]
    rows = (c.scenery.lines.starts.long()[:, None] + torch.arange(af, device='cuda')[None]).flatten()   # the agents' lines
    for fields in [('distances',), ('indices',), ('indices', 'distances'), ('locations',), ('dots', 'distances'),
                   ('indices', 'locations', 'dots', 'distances')]:
        c.scenery.lines.vals[rows] = 0.                                       # the draw step must still happen
        part = cuda.render(c.scenery, c.agents, fields=fields)
        for f in cuda.FIELDS:
            if f in fields:
                assert np.array_equal(_bits(getattr(part, f)), _bits(getattr(full, f))), (fields, f)
            else:
                assert getattr(part, f) is None
        assert torch.equal(c.scenery.lines.vals, lines_after)
    if res % 4 == 0:
        with_rgb = cuda.render(c.scenery, c.agents, fields=(), pooled=dict(subsample=4, max_depth=7., centre=True)) if grid or n_agents == 1 else None
        lean = cuda.render(c.scenery, c.agents, fields=('distances',), pooled=dict(subsample=4, max_depth=7., rgb=False, centre=True))
        assert lean.obs_rgb is None and lean.obs_depth.shape == (6, n_agents, res//4)
        want = (1 - ((full.distances - c.agent_radius)/7.).clamp(0, 1)).view(6, n_agents, res//4, 4).mean(-1)
        torch.testing.assert_close(lean.obs_depth, want, rtol=0, atol=1e-6)
        idx = full.indices.view(6, n_agents, res//4, 4)[..., 2]              # the middle ray of a pixel (deathmatch.py:74-80)
        mid = idx[..., [res//8 - 1, res//8]]
        af = n_agents*c.scenery.model.shape[0]
        want_c = torch.where((mid >= 0) & (mid < af), mid//c.scenery.model.shape[0], torch.full_like(mid, -1))
        assert torch.equal(lean.obs_centre, want_c)
        if with_rgb is not None:
            assert torch.equal(with_rgb.obs_depth, lean.obs_depth) and torch.equal(with_rgb.obs_centre, lean.obs_centre)


def test_first_sight_books_without_colour_equal_those_kept_with_it():
    """`seen=` next to a depth-only request (Explorer(depth_only=True)): the same stamps and tallies as next to RGB-D."""
    from megastep_amd import core, cubicasa, cuda, modules, scene
    from megastep_amd.demo.envs import explorer
    np.random.seed(8); torch.manual_seed(8)
    gs = cubicasa.sample(12, n_unique=16)
    c = core.Core(scene.scenery(gs, 1, random=np.random.RandomState(0)), res=256, fov=130)
    modules.RandomSpawns(gs, c)(c.agent_full(True))
    a, b = explorer.SeenTexels(c.scenery, 12), explorer.SeenTexels(c.scenery, 12)
    rgb, depth = modules.RGB(c, subsample=4), modules.Depth(c, subsample=4)
    rng = np.random.RandomState(1)
    for step in range(6):
        util.random_velocities(c, rng)
        cuda.physics(c.scenery, c.agents)
        with_colour = modules.render(c, observers=(rgb, depth), fields=(), seen=a.books)
        without = modules.render(c, observers=(depth,), fields=(), seen=b.books)
        assert torch.equal(a.stamp, b.stamp) and torch.equal(a.tally, b.tally)
        assert torch.equal(with_colour.pooled_depth, without.pooled_depth)
    assert a.tally.min() > 20


def test_depth_only_explorer():
    from megastep_amd import arrdict, cubicasa
    from megastep_amd.demo import Explorer
    np.random.seed(3); torch.manual_seed(3)
    geometries = cubicasa.sample(16, n_unique=16)
    env = Explorer(16, geometries=geometries, depth_only=True)
    w = env.reset()
    assert set(w.obs.keys()) == {'d', 'imu'} and w.obs.d.shape == (16, 1, 1, 1, 64) and 'rgb' not in env.obs_space
    for _ in range(5):
        acts = arrdict.arrdict(actions=torch.randint(0, 7, (16, 1), device='cuda'))
        w = env.step(acts)
    assert torch.isfinite(w.obs.d).all() and 0 < float(w.obs.d.mean()) < 1 and (w.reward >= 0).all()
