"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on identical seeded inputs.

Bar (BASELINE.json north_star): bit-exact collision masks and hit indices; <= 1e-5 on float positions / pixels."""
import os

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _world(n_envs, n_agents, res, fov, seed=0, large=False, toy=None):
    from megastep_amd import core, cubicasa, scene, toys
    np.random.seed(seed)
    if toy is not None:
        geometries = n_envs*[getattr(toys, toy)()]
    else:
        geometries = cubicasa.sample(n_envs, n_unique=64, seed=seed + 1, large=large)
    scenery = scene.scenery(geometries, n_agents, device='cuda', random=np.random.RandomState(seed))
    c = core.Core(scenery, res=res, fov=fov, fps=10)
    util.spawn(c, geometries, seed=seed)
    return c, geometries


@pytest.mark.parametrize('n_envs,n_agents,res,fov,toy', [
    (1, 1, 8, 130, 'box'),
    (3, 1, 64, 130, 'box'),
    (2, 2, 64, 70, 'column'),
    (8, 1, 64, 130, None),
    (8, 4, 64, 130, None),
    (5, 4, 128, 70, None),
    (3, 3, 100, 90, None),      # ragged ray group (100 = 64 + 36), A does not divide the wave count
    (2, 6, 32, 130, None),      # more agents than waves per workgroup
    (2, 4, 512, 70, None),      # the reference's own Deathmatch resolution
])
def test_step_matches_oracle(n_envs, n_agents, res, fov, toy):
    from megastep_amd import cuda
    c, geometries = _world(n_envs, n_agents, res, fov, toy=toy)
    ref = util.OracleWorld(c)
    np.testing.assert_allclose(c.scenery.baked.vals.cpu().numpy(), ref.bake(), rtol=0, atol=1e-5)
    ref.pull_baked(c)
    rng = np.random.RandomState(7)
    exact = []
    for step in range(4):
        util.random_velocities(c, rng, speed=4. if step % 2 else 40.)
        ref.pull_agents(c)
        p = cuda.physics(c.scenery, c.agents)
        r = cuda.render(c.scenery, c.agents)
        prog_ref, agents_ref = ref.physics()
        render_ref = ref.render()
        util.assert_physics_matches(c, p, prog_ref, agents_ref)
        util.assert_render_matches(c, r, render_ref)
        np.testing.assert_allclose(c.scenery.lines.vals.cpu().numpy(), ref.scene.lines_vals, rtol=0, atol=1e-6)
        exact.append([util.exact_fraction(p.progress.cpu().numpy(), prog_ref),
                      util.exact_fraction(r.distances.cpu().numpy(), render_ref['distances']),
                      util.exact_fraction(r.screen.cpu().numpy(), render_ref['screen'])])
    print('bitwise-equal fractions (progress, distances, screen):', np.mean(exact, 0))


def test_agents_see_each_other():
    """Two agents facing each other in a box: exercises agent-line hits and the dynamic lighting path."""
    from megastep_amd import cuda
    c, _ = _world(4, 2, 64, 70, toy='box')
    c.agents.positions[:] = torch.tensor([[2.5, 3.5], [4.5, 3.5]], device=c.device)
    c.agents.angles[:] = torch.tensor([0., 180.], device=c.device)
    ref = util.OracleWorld(c)
    ref.bake(); ref.pull_baked(c); ref.pull_agents(c)
    r = cuda.render(c.scenery, c.agents)
    want = ref.render()
    idx = r.indices.cpu().numpy()
    assert ((idx >= 0) & (idx < 16)).any(), 'expected rays to land on the other agent'
    assert (r.screen.cpu().numpy()[(idx >= 0) & (idx < 16)] > 0).any()
    util.assert_render_matches(c, r, want)


def test_known_answer_from_the_docs():
    """reference: docs/tutorials/minimal-env/index.rst:140-145 - box(5), agent at (3, 3), velocity (1000, 0)."""
    from megastep_amd import cuda
    c, _ = _world(128, 1, 64, 130, toy='box')
    c.agents.positions[:] = torch.as_tensor([3., 3.], device=c.device)
    c.agents.angles[:] = 0.
    c.agents.velocity[:] = torch.as_tensor([1000., 0.], device=c.device)
    p = cuda.physics(c.scenery, c.agents)
    np.testing.assert_allclose(c.agents.positions.cpu().numpy(), np.tile([5.8649, 3.0], (128, 1, 1)), atol=5e-5)
    assert (p.progress.cpu().numpy() < 1).all()
    assert (c.agents.velocity.cpu().numpy() == 0).all()


def test_errors_are_loud():
    from megastep_amd import cuda, core, scene, toys
    scenery = scene.scenery([toys.box()], 1, device='cpu', bake=False)
    with pytest.raises(RuntimeError, match='GPU'):
        cuda.bake(scenery)
    agents = core._init_agents(1, 1, 'cpu')
    with pytest.raises(RuntimeError, match='GPU'):
        cuda.physics(scenery, agents)


def _custom_world(walls_per_env, n_agents, res, fov, positions, angles, lights=None):
    """A Core over hand-made wall sets (one (W, 2, 2) array per env), agents placed explicitly."""
    from megastep_amd import core, scene, arrdict
    geoms = [arrdict.arrdict(walls=np.asarray(w, float), lights=np.array([[2., 2.]]) if lights is None else lights,
                             masks=np.ones((4, 4), np.int16), res=.2) for w in walls_per_env]
    np.random.seed(0)
    scenery = scene.scenery(geoms, n_agents, device='cuda', random=np.random.RandomState(0))
    c = core.Core(scenery, res=res, fov=fov, fps=10)
    c.agents.positions[:] = torch.as_tensor(np.asarray(positions, np.float32), device=c.device)
    c.agents.angles[:] = torch.as_tensor(np.asarray(angles, np.float32), device=c.device)
    return c


def test_hysteresis_band_adversarial():
    """Stacks of (nearly) coincident walls in shuffled order: the 1e-4 z-fight rule (kernels.cu:369) makes the
    winner depend on line order, which is exactly what the atomic-argmin path has to reproduce. Offsets straddle
    the band: 0, 2e-5, 5e-5, 9e-5, 1e-4, 1.1e-4, 2e-4, 1e-3."""
    rng = np.random.RandomState(3)
    offsets = np.array([0., 0., 2e-5, 5e-5, 9e-5, 1e-4, 1.1e-4, 2e-4, 3e-4, 1e-3])
    envs, pos, ang = [], [], []
    for e in range(48):
        k = rng.randint(2, 9)
        xs = 4. + rng.choice(offsets, k)*rng.choice([1, 1, -1], k) + rng.choice([0., 0., .5], k)
        walls = [[[x, 1. + rng.uniform(-.2, .2)], [x, 3. + rng.uniform(-.2, .2)]] for x in xs]
        if e % 3 == 0:      # an enclosing corner so that rays through shared vertices occur too
            walls += [[[4., 3.], [2., 3.]], [[2., 3.], [2., 1.]], [[2., 1.], [4., 1.]]]
        if e % 4 == 0:      # duplicates and reversed duplicates
            walls += [walls[0], [walls[1][1], walls[1][0]]]
        order = rng.permutation(len(walls))
        envs.append(np.array(walls)[order])
        pos.append([[rng.uniform(2.2, 3.9), rng.uniform(1.5, 2.5)]])
        ang.append([rng.uniform(-30, 30)])
    # pad every env to its own length is fine: Ragged
    c = _custom_world(envs, 1, 64, 100, pos, ang)
    from megastep_amd import cuda
    ref = util.OracleWorld(c)
    ref.bake(); ref.pull_baked(c); ref.pull_agents(c)
    r = cuda.render(c.scenery, c.agents, telemetry=True)
    util.assert_render_matches(c, r, ref.render())
    if not os.environ.get('MEGASTEP_RENDER_IMPL', '').startswith('s'):
        # the scenes are there to drive the sequential-fold fallbacks: make sure they did (render_prep_kernel's telemetry)
        _, folded_rays, lane_parallel_waves = r._telemetry[:3].tolist()
        assert folded_rays > 100 and lane_parallel_waves > 5, (folded_rays, lane_parallel_waves)


def test_agent_wedged_between_coincident_walls():
    """An agent 0.11 m from a wall that is doubled by a flush pillar face: most of its rays tie."""
    wall = [[[2., 1.], [2., 2.2]], [[2., 2.2], [2., 4.]]]
    pillar = [[[2., 2.], [2.3, 2.]], [[2.3, 2.], [2.3, 2.4]], [[2.3, 2.4], [2., 2.4]], [[2., 2.4], [2., 2.]]]
    triple = pillar + [[[2., 2.05], [2., 2.35]]]
    box = [[[1., 1.], [4., 1.]], [[4., 1.], [4., 4.]], [[4., 4.], [1., 4.]], [[1., 4.], [1., 1.]]]
    envs = [np.array(wall + pillar + box), np.array(pillar + wall + box), np.array(wall + triple + box),
            np.array(triple[::-1] + wall + box)]
    c = _custom_world(envs, 1, 64, 130, [[[2.11, 2.2]]]*4, [[180.], [170.], [-175.], [180.]])
    from megastep_amd import cuda
    ref = util.OracleWorld(c)
    ref.bake(); ref.pull_baked(c); ref.pull_agents(c)
    r = cuda.render(c.scenery, c.agents)
    util.assert_render_matches(c, r, ref.render())


def test_full_benchmark_size_matches_oracle():
    """BASELINE.json's metric shape - 4096 envs x 4 agents x 64 rays on synthetic cubicasa plans, the bench's own world: 1024
    distinct plans tiled (N // 4), its wall grid and light grid the size the bench steps through - against the
    oracle (OpenMP over envs), two steps."""
    from megastep_amd import cuda
    import bench
    bench.PLAN_CONTEXT = 'subprocess'                                      # (this process has been using its GPU for a while)
    c, geometries = bench.build_world(4096, 4, 64, 130., torch.device('cuda'), seed=1, n_unique=bench.plan_count(4096, 4))
    assert len({id(g) for g in geometries}) == 1024 and c.scenery.grid_report()['wall_grid']['floorplans'] == 1024
    ref = util.OracleWorld(c)
    ref.pull_baked(c)
    rng = np.random.RandomState(5)
    for step in range(2):
        util.random_velocities(c, rng, speed=6.)
        ref.pull_agents(c)
        p = cuda.physics(c.scenery, c.agents)
        r = cuda.render(c.scenery, c.agents)
        prog_ref, agents_ref = ref.physics()
        util.assert_physics_matches(c, p, prog_ref, agents_ref)
        util.assert_render_matches(c, r, ref.render())


def test_full_benchmark_size_on_oblique_floorplans_matches_oracle():
    """The headline shape on floorplans turned by seeded angles with diagonal partitions (cubicasa.sample(oblique=True), the bench's
    `shapes.headline_oblique` world): the reference's walls are exteriors of arbitrary SVG polygons (geometry.py:43-57), and until
    round 6 every plan-scale world the kernels had met - the occlusion cull's pass-1 intervals, the view arcs, the light grid's
    verdicts - was axis-aligned.  All 4096 envs against the oracle, two steps, bake included."""
    from megastep_amd import cuda
    import bench
    bench.PLAN_CONTEXT = 'subprocess'
    c, geometries = bench.build_world(4096, 4, 64, 130., torch.device('cuda'), seed=1, n_unique=bench.plan_count(4096, 4), oblique=True)
    assert len({id(g) for g in geometries}) == 1024 and c.scenery.grid_report()['wall_grid']['floorplans'] == 1024
    walls = geometries[0].walls
    d = walls[:, 1] - walls[:, 0]
    assert (np.abs(d).min(1) > 1e-3).mean() > .9, 'hardly a wall of an oblique plan is aligned with an axis'
    ref = util.OracleWorld(c)
    np.testing.assert_allclose(c.scenery.baked.vals.cpu().numpy(), ref.bake(), rtol=0, atol=1e-5)
    ref.pull_baked(c)
    rng = np.random.RandomState(6)
    for step in range(2):
        util.random_velocities(c, rng, speed=6.)
        ref.pull_agents(c)
        p = cuda.physics(c.scenery, c.agents)
        r = cuda.render(c.scenery, c.agents)
        prog_ref, agents_ref = ref.physics()
        util.assert_physics_matches(c, p, prog_ref, agents_ref)
        util.assert_render_matches(c, r, ref.render())


@pytest.mark.parametrize('n_envs,n_agents,res,fov,large', [(8, 4, 64, 130, False), (6, 1, 256, 130, True), (5, 4, 512, 70, False), (4, 2, 100, 160, False)])
def test_step_on_oblique_floorplans_matches_oracle(n_envs, n_agents, res, fov, large):
    """Small oblique worlds through every instantiation family: the plain one, wide single-agent fans on large plans, the
    reference Deathmatch's 512 rays, a ragged last group at a wide view - four steps each, lines written back included."""
    from megastep_amd import core, cubicasa, cuda, scene
    np.random.seed(11)
    geometries = cubicasa.sample(n_envs, n_unique=16, seed=12, large=large, oblique=True)
    scenery = scene.scenery(geometries, n_agents, device='cuda', random=np.random.RandomState(11))
    c = core.Core(scenery, res=res, fov=fov, fps=10)
    util.spawn(c, geometries, seed=11)
    ref = util.OracleWorld(c)
    np.testing.assert_allclose(c.scenery.baked.vals.cpu().numpy(), ref.bake(), rtol=0, atol=1e-5)
    ref.pull_baked(c)
    rng = np.random.RandomState(8)
    for step in range(4):
        util.random_velocities(c, rng, speed=4. if step % 2 else 40.)
        ref.pull_agents(c)
        p = cuda.physics(c.scenery, c.agents)
        r = cuda.render(c.scenery, c.agents)
        prog_ref, agents_ref = ref.physics()
        util.assert_physics_matches(c, p, prog_ref, agents_ref)
        util.assert_render_matches(c, r, ref.render())
        np.testing.assert_allclose(c.scenery.lines.vals.cpu().numpy(), ref.scene.lines_vals, rtol=0, atol=1e-6)


_obstructed = util.obstructed


def test_light_grid_verdicts_hold_for_every_sampled_point():
    """The grid may say LIT / DARK only where it is true for EVERY point of the cell: sample points in cells and
    evaluate the reference's obstructed() test against all walls."""
    c, _ = _world(6, 2, 64, 130, seed=3)
    sc = c.scenery
    vals, starts, geom, cell, _, lists, pool, pool_rows = (t.cpu().numpy() if torch.is_tensor(t) else t for t in sc._lg)
    pool = pool.astype(np.uint32)
    rng = np.random.RandomState(0)
    n_lit = n_dark = n_open = n_listed = n_cells = 0
    for e in range(6):
        walls = sc.lines[e].cpu().numpy()[16:]
        lights = sc.lights[e].cpu().numpy()
        ox, oy, nx, ny = geom[e]
        nx, ny = int(nx), int(ny)
        cells = rng.choice(nx*ny, 150, replace=False)
        for cidx in cells:
            ix, iy = cidx % nx, cidx // nx
            pts = (np.array([ox + ix*cell, oy + iy*cell]) + rng.uniform(0, cell, (24, 2))).astype(np.float32)
            pts = np.concatenate([pts, np.array([[ox + ix*cell, oy + iy*cell], [ox + (ix + 1)*cell, oy + (iy + 1)*cell]], np.float32)])
            words = vals[starts[e] + cidx].astype(np.uint32)
            first, header = lists[starts[e] + cidx].astype(np.uint32)
            listed = header != 0
            cands = pool[int(first):int(first) + int(header & 0x7fffffff)] if listed else []
            n_listed += int(listed); n_cells += 1
            for i, light in enumerate(lights[:64]):
                state = (words[i >> 4] >> np.uint32(2*(i & 15))) & 3
                if state == 0 and not listed:
                    continue
                blocked = _obstructed(light[:2].astype(np.float32), pts, walls).any(1)
                if state == 0:
                    # an open light: the candidate walls alone must reproduce every point's verdict
                    mine = [int(c & 0xffffff) for c in cands if (int(c) >> 24) & 63 == i]
                    assert all(int(c) >> 31 for c in cands)
                    # next to every candidate, its wall as (a, b - a): what the renderer reads instead of chasing the number
                    kept = pool_rows[int(first):int(first) + len(cands)]
                    w = walls[[int(c & 0xffffff) for c in cands]].reshape(-1, 4)
                    assert np.array_equal(kept, np.concatenate([w[:, :2], w[:, 2:] - w[:, :2]], 1))
                    few = _obstructed(light[:2].astype(np.float32), pts, walls[mine]).any(1) if mine else np.zeros(len(pts), bool)
                    assert (few == blocked).all(), (e, cidx, i, 'candidate list misses a blocker')
                    n_open += 1
                    continue
                if state == 1:
                    assert not blocked.any(), (e, cidx, i, 'LIT cell has a shadowed point')
                    n_lit += 1
                else:
                    assert blocked.all(), (e, cidx, i, 'DARK cell has a lit point')
                    n_dark += 1
    assert n_lit > 50 and n_dark > 1000 and n_open > 100, (n_lit, n_dark, n_open)
    assert n_listed > .9*n_cells, (n_listed, n_cells)          # the pool is rarely short
    assert 0 < int(pool[0]) < len(pool)


@pytest.mark.parametrize('res,fov,cell,budget', [(64, 130, .25, None), (256, 70, 1., None), (128, 100, .5, None), (64, 130, .125, None),
                                                 (64, 130, .125, 'no rows'), (128, 70, .125, 'coarser')])
def test_crowded_rooms_exercise_dynamic_lighting(monkeypatch, res, fov, cell, budget):
    """Four agents packed into one room of each plan, looking at each other: many rays land on agents, under every
    mix of lit / shadowed / partly shadowed lights.  With coarser light-grid cells more lights stay open and the
    candidate lists grow past one LDS batch (64 pairs): several lists per wave, several batches per list, several rounds
    of (ray, candidate) pairs per batch."""
    from megastep_amd import cuda
    monkeypatch.setattr(cuda.Scenery, 'LIGHT_GRID_CELL', cell)
    monkeypatch.setattr(cuda.Scenery, 'LIGHT_GRID_POOL', 12 if cell <= .25 else 200)
    if budget is not None:
        # LIGHT_GRID_BYTES: a grid over its budget goes without the candidates' rows (ms_render then fetches a
        # candidate's wall from `lines`), then on cells twice, four times the size - the same picture either way
        full = _world(24, 4, res, fov, seed=5)[0].scenery.grid_report()['light_grid']['bytes']
        monkeypatch.setattr(cuda.Scenery, 'LIGHT_GRID_BYTES', full - 1 if budget == 'no rows' else full//12)
    c, geometries = _world(24, 4, res, fov, seed=5)
    if budget is not None:
        rep = c.scenery.grid_report()['light_grid']
        assert not rep['candidate_rows'] and c.scenery._lg[7] is None and (rep['cell'] == cell) == (budget == 'no rows')
    rng = np.random.RandomState(2)
    pos = np.zeros((24, 4, 2), np.float32)
    ang = np.zeros((24, 4), np.float32)
    from megastep_amd import geometry
    for e, g in enumerate(geometries):
        room = rng.randint(1, g['masks'].max() + 1)
        free = np.stack((g['masks'] == room).nonzero(), -1)
        centre = geometry.centers(free[rng.randint(len(free))], g['masks'].shape, g['res'])
        pos[e] = centre + rng.uniform(-.45, .45, (4, 2))
        ang[e] = np.degrees(np.arctan2(*(pos[e].mean(0) - pos[e]).T[::-1])) + rng.uniform(-20, 20, 4)
    c.agents.positions[:] = torch.as_tensor(pos, device=c.device)
    c.agents.angles[:] = torch.as_tensor(ang, device=c.device)
    ref = util.OracleWorld(c)
    ref.pull_baked(c); ref.pull_agents(c)
    r = cuda.render(c.scenery, c.agents)
    idx = r.indices.cpu().numpy()
    assert ((idx >= 0) & (idx < 32)).mean() > .05, 'expected plenty of rays on agents'
    if cell > .25:
        counts = (c.scenery._lg[5].view(-1, 2)[:, 1].cpu().numpy().astype(np.int64)) & 0x7fffffff
        assert counts.max() > 64, 'expected candidate lists longer than one batch'
    util.assert_render_matches(c, r, ref.render())


@pytest.mark.parametrize('n_envs,n_agents,res,fov', [(3, 1, 256, 130), (2, 2, 64, 170), (2, 1, 1, 90)])
def test_large_maps_and_extreme_views(n_envs, n_agents, res, fov):
    """800-1200 wall plans (BASELINE config 5's geometry), a 170 degree fan, a single ray."""
    from megastep_amd import cuda
    c, _ = _world(n_envs, n_agents, res, fov, seed=9, large=True)
    ref = util.OracleWorld(c)
    np.testing.assert_allclose(c.scenery.baked.vals.cpu().numpy(), ref.bake(), rtol=0, atol=1e-5)
    ref.pull_baked(c)
    rng = np.random.RandomState(3)
    for _ in range(2):
        util.random_velocities(c, rng, speed=8.)
        ref.pull_agents(c)
        p = cuda.physics(c.scenery, c.agents)
        r = cuda.render(c.scenery, c.agents)
        util.assert_physics_matches(c, p, *ref.physics())
        util.assert_render_matches(c, r, ref.render())


def test_ragged_edge_cases():
    """An env without lights, an env with a single wall, an env with more walls than one LDS staging pass of the
    bake kernels holds (2048), side by side with a normal one."""
    from megastep_amd import cuda, toys, arrdict
    rng = np.random.RandomState(0)
    box = toys.box()
    many = np.concatenate([box.walls] + [np.array([[[x, y], [x + .03, y + .02]]]) for x in np.linspace(1.5, 5.5, 50) for y in np.linspace(1.5, 5.5, 44)])
    assert len(many) > 2048
    geoms = [
        arrdict.arrdict(walls=box.walls, lights=np.zeros((0, 2)), masks=box.masks, res=.2),          # dark room
        arrdict.arrdict(walls=box.walls[:1], lights=box.lights, masks=box.masks, res=.2),            # one wall
        arrdict.arrdict(walls=many, lights=np.array([[3.5, 3.5], [2., 5.]]), masks=box.masks, res=.2),
        box]
    from megastep_amd import core, scene
    np.random.seed(0)
    scenery = scene.scenery(geoms, 2, device='cuda', random=np.random.RandomState(0))
    c = core.Core(scenery, res=64, fov=130)
    c.agents.positions[:] = torch.as_tensor(rng.uniform(2, 5, (4, 2, 2)).astype(np.float32), device=c.device)
    c.agents.angles[:] = torch.as_tensor(rng.uniform(-180, 180, (4, 2)).astype(np.float32), device=c.device)
    ref = util.OracleWorld(c)
    np.testing.assert_allclose(scenery.baked.vals.cpu().numpy(), ref.bake(), rtol=0, atol=1e-5)
    ref.pull_baked(c)
    for _ in range(2):
        util.random_velocities(c, rng, speed=5.)
        ref.pull_agents(c)
        p = cuda.physics(c.scenery, c.agents)
        r = cuda.render(c.scenery, c.agents)
        util.assert_physics_matches(c, p, *ref.physics())
        util.assert_render_matches(c, r, ref.render())
    assert (r.screen[0][r.indices[0] >= 16] > 0).any()        # ambient light only, but not black


def test_render_is_deterministic_and_leaves_inputs_alone():
    from megastep_amd import cuda
    c, _ = _world(16, 4, 64, 130, seed=4)
    before = {k: getattr(c.agents, k).clone() for k in ('angles', 'positions', 'angvelocity', 'velocity')}
    a = cuda.render(c.scenery, c.agents)
    b = cuda.render(c.scenery, c.agents)
    for k in ('indices', 'locations', 'dots', 'distances', 'screen'):
        x, y = getattr(a, k), getattr(b, k)
        assert torch.equal(torch.nan_to_num(x.float(), nan=-7.), torch.nan_to_num(y.float(), nan=-7.)), k
    for k, v in before.items():
        assert torch.equal(getattr(c.agents, k), v), k


def test_unbaked_scenery_and_shards_render_exactly():
    """A Scenery that was never baked has an all-unknown light grid and baked == 1: render must still be exact.
    A shard cut from a baked scenery carries baked lighting and light grid over."""
    from megastep_amd import core, cuda, cubicasa, scene, sharding
    np.random.seed(0)
    gs = cubicasa.sample(6, n_unique=16, seed=3)
    raw = scene.scenery(gs, 3, device='cuda', random=np.random.RandomState(0), bake=False)
    for sc, geoms in [(raw, gs)]:
        c = core.Core(sc, res=64, fov=130)
        util.spawn(c, geoms, seed=1)
        c.agents.positions[:, 1] = c.agents.positions[:, 0] + torch.tensor([.4, .1], device=c.device)     # someone to look at
        ref = util.OracleWorld(c)
        ref.pull_agents(c)
        util.assert_render_matches(c, cuda.render(c.scenery, c.agents), ref.render())
    full = scene.scenery(gs, 3, device='cuda', random=np.random.RandomState(0))
    shard = sharding.shard_scenery(full, 1, 2)
    assert torch.equal(shard._lg[0][:-1], full._lg[0][int(full._lg[1][3]):-1]) and shard._lg[0].any()     # (both end in a padding row)
    c = core.Core(shard, res=64, fov=130)
    util.spawn(c, gs[3:], seed=2)
    c.agents.positions[:, 2] = c.agents.positions[:, 0] + torch.tensor([.1, .45], device=c.device)
    ref = util.OracleWorld(c)
    ref.pull_agents(c)
    util.assert_render_matches(c, cuda.render(c.scenery, c.agents), ref.render())


def test_more_than_64_lights_and_agents(monkeypatch):
    """An env with more lights than the light grid holds per env (64) gets no cells of it, and the rays that land on an
    agent there meet the walls group of 64 lights after group of 64 - inside the render kernel, next to envs that do have
    their grids: one launch, pooled observations and all (reference: kernels.cu:245-267 has no such limit). Without any
    grid the separate lighting kernel does the same. Past 64 agents per env the per-wave agent cache is bypassed."""
    from megastep_amd import arrdict, core, cuda, modules, scene, toys
    rng = np.random.RandomState(0)
    box = toys.box()
    pillars = np.concatenate([np.array([[[x, y], [x + .2, y]], [[x + .2, y], [x + .2, y + .2]], [[x + .2, y + .2], [x, y + .2]],
                                        [[x, y + .2], [x, y]]]) for x, y in rng.uniform(1.5, 5.3, (6, 2))])
    walls = np.concatenate([box.walls, pillars])
    geoms = [arrdict.arrdict(walls=walls, lights=rng.uniform(1.2, 5.8, (k, 2)), masks=box.masks, res=.2) for k in (70, 9, 150, 64)]
    for grid in (True, False):
        monkeypatch.setattr(cuda.Scenery, 'LIGHT_GRID', grid)
        np.random.seed(0)
        sc = scene.scenery(geoms, 3, device='cuda', random=np.random.RandomState(0))
        assert (sc._as_struct().lg_vals is not None) == grid
        if grid:                                                    # cells for the envs the grid can hold, none for the others
            assert (sc._lg[2][:, 2] > 0).tolist() == [False, True, False, True]
        c = core.Core(sc, res=64, fov=130)
        spots = np.array([[3., 3.], [4., 3.1], [3.5, 4.]], np.float32) + rng.uniform(-.2, .2, (4, 3, 2)).astype(np.float32)
        c.agents.positions[:] = torch.as_tensor(spots, device=c.device)     # a triangle of agents looking at each other
        c.agents.angles[:] = torch.as_tensor(np.array([30., 150., -90.], np.float32) + rng.uniform(-10, 10, (4, 3)).astype(np.float32), device=c.device)
        ref = util.OracleWorld(c)
        np.testing.assert_allclose(sc.baked.vals.cpu().numpy(), ref.bake(), rtol=0, atol=1e-5)
        ref.pull_baked(c); ref.pull_agents(c)
        r = cuda.render(c.scenery, c.agents)
        lit = (r.indices >= 0) & (r.indices < 24)
        assert all(lit[e].any() for e in range(4)), 'rays should land on agents in every env'
        want = ref.render()
        util.assert_render_matches(c, r, want)
        if grid:                                                    # ... and the fused observations stay available
            rgb, depth = modules.RGB(c, subsample=4), modules.Depth(c, subsample=4)
            frame = modules.render(c, observers=(rgb, depth), fields=(), centre=True)
            assert 'pooled_rgb' in frame and 'screen' not in frame
            full = torch.as_tensor(want['screen'], device=c.device).permute(0, 1, 3, 2).unsqueeze(3)
            torch.testing.assert_close(rgb(frame), modules.downsample(full, 4).mean(-1), rtol=0, atol=1e-5)
    monkeypatch.setattr(cuda.Scenery, 'LIGHT_GRID', True)

    crowd = scene.scenery([toys.box(8)], 66, device='cuda', random=np.random.RandomState(0))
    c = core.Core(crowd, res=16, fov=130)
    c.agents.positions[:] = torch.as_tensor(rng.uniform(1.5, 8.5, (1, 66, 2)).astype(np.float32), device=c.device)
    c.agents.angles[:] = torch.as_tensor(rng.uniform(-180, 180, (1, 66)).astype(np.float32), device=c.device)
    ref = util.OracleWorld(c)
    ref.bake(); ref.pull_baked(c)
    util.random_velocities(c, rng, speed=3.)
    ref.pull_agents(c)
    p = cuda.physics(c.scenery, c.agents)
    r = cuda.render(c.scenery, c.agents)
    util.assert_physics_matches(c, p, *ref.physics())
    util.assert_render_matches(c, r, ref.render())


@pytest.mark.parametrize('n_agents,res,subsample', [(3, 256, 4), (4, 64, 1), (2, 128, 8), (1, 96, 2), (2, 512, 4)])
def test_fused_observations_match_the_host_modules(n_agents, res, subsample):
    """The pooled RGB-D the render kernel writes itself against the reference's chain of tensor ops
    (modules.py:138-145,170-184,211-224) on the full-resolution outputs of the same scene - and against the oracle's
    render run through the same modules on the CPU."""
    from megastep_amd import cuda, modules
    c, _ = _world(6, n_agents, res, 130, seed=11)
    rgb, depth = modules.RGB(c, subsample=subsample), modules.Depth(c, subsample=subsample, max_depth=7.)
    full = modules.render(c)
    want_rgb, want_d = rgb(full).clone(), depth(full).clone()
    fused = modules.render(c, observers=(rgb, depth), fields=('indices',))
    assert set(fused.keys()) >= {'indices', 'pooled_rgb', 'pooled_depth'} and 'screen' not in fused and 'distances' not in fused
    got_rgb, got_d = rgb(fused), depth(fused)
    assert got_rgb.shape == want_rgb.shape == (6, n_agents, 3, 1, res//subsample)
    assert got_d.shape == want_d.shape == (6, n_agents, 1, 1, res//subsample)
    assert torch.equal(fused.indices, full.indices)
    torch.testing.assert_close(got_rgb, want_rgb, rtol=0, atol=1e-6)
    torch.testing.assert_close(got_d, want_d, rtol=0, atol=1e-6)
    # the oracle's render through the same modules, on the CPU
    ref = util.OracleWorld(c)
    ref.pull_baked(c); ref.pull_agents(c)
    o = ref.render()
    class CpuCore: agent_radius, res = c.agent_radius, c.res
    from megastep_amd import arrdict
    r_cpu = arrdict.arrdict(distances=torch.as_tensor(o['distances']).unsqueeze(2),
                            screen=torch.as_tensor(o['screen']).unsqueeze(2).permute(0, 1, 4, 2, 3))
    cpu_rgb = modules.RGB(CpuCore, n_agents=n_agents, subsample=subsample)(r_cpu)
    cpu_d = modules.Depth(CpuCore, n_agents=n_agents, subsample=subsample, max_depth=7.)(r_cpu)
    np.testing.assert_allclose(got_rgb.cpu().numpy(), cpu_rgb.numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(got_d.cpu().numpy(), cpu_d.numpy(), rtol=0, atol=1e-5)
    assert float(got_rgb.max()) > 0 and 0 < float(got_d.mean()) < 1


def test_unwanted_outputs_are_skipped_and_wanted_ones_unchanged():
    from megastep_amd import cuda
    c, _ = _world(5, 3, 100, 90, seed=4)
    full = cuda.render(c.scenery, c.agents)
    for fields in [('indices',), ('distances', 'screen'), ('locations', 'dots'), cuda.FIELDS]:
        part = cuda.render(c.scenery, c.agents, fields=fields)
        for f in cuda.FIELDS:
            if f in fields:
                assert torch.equal(getattr(part, f), getattr(full, f), ) or \
                    torch.equal(torch.nan_to_num(getattr(part, f), nan=-7.), torch.nan_to_num(getattr(full, f), nan=-7.)), f
            else:
                assert getattr(part, f) is None
    only_obs = cuda.render(c.scenery, c.agents, fields=(), pooled=dict(subsample=4, max_depth=10.))
    assert only_obs.indices is None and only_obs.obs_rgb.shape == (5, 3, 3, 25) and only_obs.obs_depth.shape == (5, 3, 25)
    with pytest.raises(RuntimeError):
        cuda.render(c.scenery, c.agents, fields=())                              # nothing asked for at all
    with pytest.raises(RuntimeError):
        cuda.render(c.scenery, c.agents, pooled=dict(subsample=3))               # not a power of two
    with pytest.raises(RuntimeError):
        cuda.render(c.scenery, c.agents, pooled=dict(subsample=8))               # does not divide 100


def test_heading_cache_is_used_only_while_it_is_true():
    """ms_physics leaves every agent's sin/cos for the next ms_render (MsAgents.headings); an agent the caller turns in
    between must be rendered from its new angle. Same bits as the self-contained path in every case."""
    from megastep_amd import cuda
    c, _ = _world(10, 4, 64, 130, seed=13)
    rng = np.random.RandomState(5)
    util.random_velocities(c, rng, speed=3.)
    assert not c.agents._cached
    cuda.physics(c.scenery, c.agents)
    assert c.agents._cached
    h = c.agents._headings
    assert torch.equal(h[..., 0], c.agents.angles)
    s, co = h[..., 1].double(), h[..., 2].double()
    assert float((s*s + co*co - 1).abs().max()) < 1e-6
    def same(a, b):
        return all(torch.equal(torch.nan_to_num(getattr(a, f).float(), nan=-7.), torch.nan_to_num(getattr(b, f).float(), nan=-7.)) for f in cuda.FIELDS)
    assert same(cuda.render(c.scenery, c.agents), cuda.render(c.scenery, c.agents, telemetry=True))
    # turn some agents behind the cache's back (what a respawn does)
    c.agents.angles[::2, 1] += 33.
    c.agents.angles[3, :] = torch.tensor([0., -180., 179.99999, 90.], device='cuda')
    fresh = cuda.render(c.scenery, c.agents)
    assert same(fresh, cuda.render(c.scenery, c.agents, telemetry=True))
    ref = util.OracleWorld(c)
    ref.pull_baked(c); ref.pull_agents(c)
    util.assert_render_matches(c, fresh, ref.render())


def test_outputs_can_be_reused_between_calls():
    """cuda.render / cuda.physics write into the tensors of an earlier call's result when given it as `out` (the
    benchmark's hot path does, so that nothing is allocated per step): same values as fresh calls, same objects."""
    from megastep_amd import cuda
    c, _ = _world(6, 3, 64, 110, seed=4)
    rng = np.random.RandomState(0)
    held_p = held_r = None
    for step in range(3):
        util.random_velocities(c, rng)
        twin = cuda.Agents(*(t.clone() for t in (c.agents.angles, c.agents.positions, c.agents.angvelocity, c.agents.velocity)))
        p_fresh = cuda.physics(c.scenery, twin)
        r_fresh = cuda.render(c.scenery, twin)
        p = cuda.physics(c.scenery, c.agents, out=held_p)
        r = cuda.render(c.scenery, c.agents, out=held_r)
        assert held_p is None or (p is held_p and r is held_r)
        held_p, held_r = p, r
        assert torch.equal(p.progress, p_fresh.progress)
        for k in cuda.FIELDS:
            a, b = getattr(r, k), getattr(r_fresh, k)
            assert torch.equal(torch.nan_to_num(a.float(), nan=-7.), torch.nan_to_num(b.float(), nan=-7.)), k
    with pytest.raises(RuntimeError, match='same shapes'):
        cuda.render(c.scenery, c.agents, fields=('indices',), out=held_r)


def test_steps_replayed_as_a_hip_graph_equal_the_same_steps_launched_one_by_one():
    """bench.py's headline leg records K physics + render steps into one HIP graph and replays it. Same kernels, same
    arguments, stream order kept by the graph's edges: the state after the replay and the last frame must be the
    eager run's, bit for bit."""
    from megastep_amd import cuda
    c, _ = _world(96, 4, 64, 130, seed=5)
    K = 6
    g = torch.Generator(device='cuda').manual_seed(0)
    vel = 6*torch.rand((K, 96, 4, 2), device='cuda', generator=g) - 3
    angvel = 360*torch.rand((K, 96, 4), device='cuda', generator=g) - 180
    start = [t.clone() for t in (c.agents.angles, c.agents.positions)]
    inputs = (vel.clone(), angvel.clone())                       # physics zeroes the velocities of agents that collide

    def rewind():
        c.agents.angles.copy_(start[0]); c.agents.positions.copy_(start[1])
        vel.copy_(inputs[0]); angvel.copy_(inputs[1])
    views = [cuda.Agents(c.agents.angles, c.agents.positions, angvel[i], vel[i]) for i in range(K)]
    state = {}

    def step(i):
        state['p'] = cuda.physics(c.scenery, views[i], out=state.get('p'))
        state['r'] = cuda.render(c.scenery, views[i], out=state.get('r'))
    for i in range(K):
        step(i)
    want = [t.clone() for t in (c.agents.angles, c.agents.positions, vel, state['p'].progress, state['r'].indices, state['r'].screen, c.scenery.lines.vals)]
    assert (want[3] < 1).any() and (want[4] >= 0).any()
    rewind()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(K):
            step(i)
    for _ in range(2):                                           # (capture does not run anything; replay twice from the start state)
        rewind()
        graph.replay()
        torch.cuda.synchronize()
        got = (c.agents.angles, c.agents.positions, vel, state['p'].progress, state['r'].indices, state['r'].screen, c.scenery.lines.vals)
        for a, b in zip(got, want):
            assert torch.equal(torch.nan_to_num(a.float(), nan=-7.), torch.nan_to_num(b.float(), nan=-7.))


def test_more_than_65536_walls_in_an_env():
    """An env too large for the wall grid's 16-bit wall numbers gets no grid cells and meets its walls one after the other -
    all of them: the 70 000th like the 7th (the queue the renderer feeds its batches from once kept 16-bit entries for this
    path too, and walls past the 65 536th came back as their numbers modulo 65 536)."""
    from megastep_amd import cuda
    rng = np.random.RandomState(0)
    n = 70_000
    far = rng.uniform(40, 90, (n, 1, 2)) + np.concatenate([np.zeros((n, 1, 2)), rng.uniform(-.05, .05, (n, 1, 2))], 1)   # specks, far away
    room = np.array([[[1., 1.], [5., 1.]], [[5., 1.], [5., 5.]], [[5., 5.], [1., 5.]], [[1., 5.], [1., 1.]]])
    walls = np.concatenate([far[:66_000], room[:2], far[66_000:], room[2:]])
    c = _custom_world([walls, room], 1, 64, 130, [[[3., 3.]], [[2., 2.]]], [[20.], [100.]])
    assert c.scenery.lines.widths.tolist() == [n + 4 + 8, 4 + 8]
    ref = util.OracleWorld(c)
    ref.pull_baked(c)
    for angle in (20., 200.):
        c.agents.angles[0] = angle
        ref.pull_agents(c)
        r = cuda.render(c.scenery, c.agents)
        util.assert_render_matches(c, r, ref.render())
        assert int(r.indices[0].max()) > 66_000
    util.random_velocities(c, rng, speed=30.)
    ref.pull_agents(c)
    p = cuda.physics(c.scenery, c.agents)
    util.assert_physics_matches(c, p, *ref.physics())
