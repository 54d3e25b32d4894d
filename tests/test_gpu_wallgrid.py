"""The wall grid on the GPU (include/megastep_hip.h, MsScenery.wg_*): the lists the gfx950 scan builds are the ones its
host instantiation builds (whose exactness tests/test_wallgrid.py establishes against the oracle), and the kernels that
walk them leave the same bits as the kernels that meet every wall - and as the oracle - also where the lists do not
apply (agents outside the grid, faster than the near lists allow, views wider than the vis lists were built for)."""
import numpy as np
import pytest
import torch

from tests import util
from tests.test_wallgrid import scan_cell, wall_arc, NEAR, REACH, REACH_LO

pytestmark = pytest.mark.gpu


def _world(n_envs, n_agents, res=64, fov=130., seed=0, n_unique=8, large=False, grid=True):
    from megastep_amd import core, cubicasa, cuda, scene
    np.random.seed(seed)
    pool = cubicasa.sample(n_unique, n_unique=64, seed=seed + 1, large=large)
    geometries = [pool[i % len(pool)] for i in range(n_envs)]
    scenery = scene.scenery(geometries, n_agents, device='cuda', random=np.random.RandomState(seed), bake=False)
    cuda.bake(scenery, wall_grid=grid)
    c = core.Core(scenery, res=res, fov=fov, fps=10)
    util.spawn(c, geometries, seed=seed)
    return c, geometries


def test_lists_built_on_the_gpu_are_the_host_scans():
    c, geometries = _world(12, 2, n_unique=5)
    sc = c.scenery
    assert sc._wg is not None
    cells, starts, geom, cell, reach_lo, reach, near, pool, rows, pool_base = sc._wg
    assert (reach_lo, reach, near) == tuple(np.float32([REACH_LO, REACH, NEAR]).astype(float)) or (reach_lo, reach, near) == (REACH_LO, REACH, NEAR)
    cells = cells.cpu().numpy().view(np.uint32)
    starts, geom, pool, rows = starts.cpu().numpy(), geom.cpu().numpy(), pool.cpu().numpy().view(np.uint32), rows.cpu().numpy()
    pool_base = pool_base.cpu().numpy()                                 # (int64: where each env's floorplan's vis lists start)
    assert pool_base[0] == pool_base[5] == pool_base[10] and len(set(pool_base[:5])) == 5 and pool_base.min() == 0
    AF = sc.n_agents*sc.model.shape[0]
    rng = np.random.RandomState(0)
    # envs of one floorplan share their cells
    assert starts[0] == starts[5] == starts[10] and starts[1] == starts[6] and len(set(starts[:5])) == 5
    fractions, excess = [], []
    for n in (0, 3, 7):
        walls = sc.lines[n].cpu().numpy()[AF:]
        origin, dims = geom[n, :2], geom[n, 2:].astype(int)
        for c_ in rng.choice(dims[0]*dims[1], 30, replace=False):
            vis, close = scan_cell(walls, origin, dims, c_, cell)
            v0, vn, n0, nn = cells[starts[n] + c_]
            v0 = int(v0) + int(pool_base[n])                            # a cell's first vis entry counts from its floorplan's base
            n_lo, n_all = nn & 0xffff, nn >> 16
            # built through the coarse level: nothing the one-level scan lists may be missing, and next to nothing more
            # (an occluder the coarse cell's list lacks has a stand-in on it, which the thresholds may judge differently)
            got, arcs = pool[v0:v0 + vn] & 0xffff, pool[v0:v0 + vn] >> 16
            for e_, arc in zip(got[::7], arcs[::7]):                      # every entry carries its wall's view arc from this cell
                assert (arc & 255, arc >> 8) == wall_arc(walls[e_], origin, dims, c_, cell)
            assert set(np.nonzero(vis)[0]) <= set(got) and len(got) <= 1.03*vis.sum() + 2 and (np.diff(got.astype(int)) > 0).all()
            excess.append(len(got) - vis.sum())
            want = np.concatenate([walls[close == 2], walls[close == 1]]).reshape(-1, 4)
            assert (n_lo, n_all) == ((close == 2).sum(), (close > 0).sum())
            np.testing.assert_array_equal(rows[n0:n0 + n_all], want)
            fractions.append(vn/len(walls))
    print('mean listed fraction:', np.mean(fractions), 'walls listed beyond the one-level scan:', np.sum(excess))
    assert np.mean(fractions) < .6


@pytest.mark.parametrize('n_agents,res,fov,large', [(4, 64, 130., False), (1, 256, 130., True), (3, 128, 70., False), (2, 64, 160., False),
                                                    (12, 64, 130., False)])   # (twelve agents: hundreds of (wall, agent) pairs an env - physics culls, compacts, then tests)
def test_kernels_with_lists_leave_the_bits_of_kernels_without(n_agents, res, fov, large):
    from megastep_amd import cuda
    worlds = [_world(96, n_agents, res, fov, seed=3, n_unique=12, large=large, grid=g)[0] for g in (True, False)]
    assert worlds[0].scenery._wg is not None and worlds[1].scenery._wg is None
    rng = np.random.RandomState(1)
    for step in range(4):
        vel = rng.uniform(-4, 4, (96, n_agents, 2)).astype(np.float32)
        spin = rng.uniform(-180, 180, (96, n_agents)).astype(np.float32)
        outs = []
        for c in worlds:
            c.agents.velocity[:] = torch.as_tensor(vel, device='cuda')
            c.agents.angvelocity[:] = torch.as_tensor(spin, device='cuda')
            p = cuda.physics(c.scenery, c.agents)
            r = cuda.render(c.scenery, c.agents)
            outs.append([p.progress, c.agents.positions, c.agents.angles, r.indices, r.distances, r.locations, r.dots, r.screen])
        for a, b in zip(*outs):
            assert torch.equal(torch.nan_to_num(a.float(), nan=-7., posinf=-8.), torch.nan_to_num(b.float(), nan=-7., posinf=-8.))
    assert (outs[0][0] < 1).any()


def test_where_the_lists_do_not_apply_every_wall_is_met():
    """Agents outside their grid or at NaN, agents whose step outruns the near lists, a crawling agent, a view wider
    than the vis lists were built for: all against the oracle."""
    from megastep_amd import core, cuda
    c, _ = _world(16, 3, seed=5)
    pos = c.agents.positions.clone()
    pos[0, 0] = torch.tensor([-50., 3.]); pos[1, 1] = torch.tensor([1e4, 1e4]); pos[2, 2] = float('nan')
    pos[3, 0] = pos[3, 1] + torch.tensor([.35, 0.], device='cuda')
    c.agents.positions[:] = pos
    rng = np.random.RandomState(2)
    for step, speed in enumerate((3., 40., 3.)):
        util.random_velocities(c, rng, speed=speed)
        if step == 2:
            c.agents.velocity[4:8] *= 1e-7                             # crawling: meets far walls (kernels.cu:91-107)
        ref = util.OracleWorld(c)
        p = cuda.physics(c.scenery, c.agents)
        r = cuda.render(c.scenery, c.agents)
        prog_ref, agents_ref = ref.physics()
        got, want = p.progress.cpu().numpy(), prog_ref
        np.testing.assert_array_equal(np.nan_to_num(got, nan=-7.), np.nan_to_num(want, nan=-7.))
        ref.agents = util.agents_dict(c.agents)
        util.assert_render_matches(c, r, ref.render())
    wide = core.Core(c.scenery, res=64, fov=172., fps=10)              # beyond MS_WALLGRID_MAX_FOV: no vis lists
    wide.agents.positions[:] = torch.nan_to_num(c.agents.positions, nan=3.)
    wide.agents.angles[:] = c.agents.angles
    ref = util.OracleWorld(wide)
    util.assert_render_matches(wide, cuda.render(wide.scenery, wide.agents), ref.render())


def test_a_grid_too_big_for_its_budget_is_coarsened_or_left_out(monkeypatch):
    from megastep_amd import cuda
    c, _ = _world(4, 1, n_unique=2)
    needed = 4*c.scenery._wg[7].numel() + 4*c.scenery._wg[8].numel()
    monkeypatch.setattr(cuda.Scenery, 'WALL_GRID_BYTES', needed//2)
    c, _ = _world(4, 1, n_unique=2)
    assert c.scenery._wg is not None and c.scenery._wg[3] > cuda.Scenery.WALL_GRID_CELL
    ref = util.OracleWorld(c)
    util.assert_render_matches(c, cuda.render(c.scenery, c.agents), ref.render())
    monkeypatch.setattr(cuda.Scenery, 'WALL_GRID_BYTES', 100)
    c, _ = _world(4, 1, n_unique=2)
    assert c.scenery._wg is None
    ref = util.OracleWorld(c)
    util.assert_render_matches(c, cuda.render(c.scenery, c.agents), ref.render())


def test_more_agents_than_lanes_take_the_sweep():
    """The near lists are dealt a lane per agent: an env with more than 64 agents meets its walls through the sweep, every
    agent of it - the last ones too (which a first version of the kernel forgot; the fuzz found it)."""
    from megastep_amd import core, cuda, scene, toys
    sc = scene.scenery([toys.box(8), toys.box(8)], 70, device='cuda', random=np.random.RandomState(0))
    assert sc._wg is not None
    c = core.Core(sc, res=8, fov=130)
    rng = np.random.RandomState(3)
    # everyone close to a wall of the 8 m box (walls at 1 and 9) and heading for it
    side = rng.randint(0, 4, (2, 70))
    along = rng.uniform(1.5, 8.5, (2, 70))
    near, far = np.where(side % 2 == 0, 1.3, 8.7), along
    pos = np.where((side < 2)[..., None], np.stack([near, far], -1), np.stack([far, near], -1)).astype(np.float32)
    vel = np.where((side < 2)[..., None], np.stack([np.where(side % 2 == 0, -3., 3.), rng.uniform(-1, 1, (2, 70))], -1),
                   np.stack([rng.uniform(-1, 1, (2, 70)), np.where(side % 2 == 0, -3., 3.)], -1)).astype(np.float32)
    c.agents.positions[:] = torch.as_tensor(pos, device='cuda')
    c.agents.velocity[:] = torch.as_tensor(vel, device='cuda')
    ref = util.OracleWorld(c)
    p = cuda.physics(c.scenery, c.agents)
    prog_ref, agents_ref = ref.physics()
    util.assert_physics_matches(c, p, prog_ref, agents_ref)
    assert (prog_ref[:, 64:] < 1).any() and (prog_ref < 1).mean() > .5


def test_walls_moved_after_bake_are_caught():
    """The contract the wall grid adds (cuda.render's docstring): its lists are a snapshot of the static walls. A wall
    nudged in place afterwards would be walked past silently - `Scenery.check_wall_grid()`, which MEGASTEP_CHECK_GRID=1 (as
    this suite runs) puts in front of every render and physics call, notices; baking again makes the world whole."""
    from megastep_amd import cuda
    c, _ = _world(4, 2, 64, 130., seed=3, n_unique=4)
    assert cuda.CHECK_GRID and c.scenery._wg is not None
    c.scenery.check_wall_grid()
    cuda.render(c.scenery, c.agents); cuda.physics(c.scenery, c.agents)          # the agents' own rows change: not the grid's business
    c.scenery.check_wall_grid()
    af = c.scenery.n_agents*c.scenery.model.shape[0]
    row = int(c.scenery.lines.starts[2]) + af + 5
    c.scenery.lines.vals[row, 1, 0] += .25
    with pytest.raises(RuntimeError, match='bake again'):
        c.scenery.check_wall_grid()
    with pytest.raises(RuntimeError, match='bake again'):
        cuda.render(c.scenery, c.agents)
    with pytest.raises(RuntimeError, match='bake again'):
        cuda.physics(c.scenery, c.agents)
    cuda.bake(c.scenery)
    ref = util.OracleWorld(c)
    ref.pull_baked(c); ref.pull_agents(c)
    util.assert_render_matches(c, cuda.render(c.scenery, c.agents), ref.render())
