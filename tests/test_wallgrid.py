"""The wall grid's two promises, checked on the CPU against the oracle (include/megastep_hip.h, MsScenery.wg_*;
DESIGN.md section 3.9): the library's host instantiation of its scan (the very functions the gfx950 kernel is compiled
from) decides which walls go on a cell's lists, the plain-C oracle - which knows nothing of any list - says what the
reference's raycast sees.

  vis   the oracle's render of a world reduced to a cell's vis list equals its render of the whole world, ray for ray
        and bit for bit, from poses all over the cell: the order-dependent nearest-hit fold (kernels.cu:352-377) ends
        where it would have;
  near  no wall within reach of a point of the cell is missing from the cell's near list.
"""
import ctypes as C
import zlib
import numpy as np
import pytest
from megastep_amd import _lib, cubicasa, scene, toys, core

CELL, NEAR, REACH_LO, REACH = .25, .12, .7, 1.3


def scan_cell(walls, origin, dims, c, cell=CELL):
    """vis (bool) and close (2 within the short reach, 1 within the long one, else 0) over the walls of cell c, from the
    library's host instantiation of its scan."""
    w = np.ascontiguousarray(walls.reshape(-1, 4), np.float32)
    vis, close = np.zeros(len(w), np.uint8), np.zeros(len(w), np.uint8)
    _lib.lib().ms_host_wallgrid_cell(w.ctypes.data, len(w), float(origin[0]), float(origin[1]), int(dims[0]), int(dims[1]),
                                     cell, int(c), NEAR, REACH_LO, REACH, vis.ctypes.data, close.ctypes.data)
    return vis.astype(bool), close


def wall_arc(wall, origin, dims, c, cell=CELL, slack=.01):
    """(first step, last step) of the run of directions the library says `wall` can be seen in from cell c."""
    x0, y0 = origin[0] + (c % dims[0])*cell - slack, origin[1] + (c//dims[0])*cell - slack
    lo, hi = C.c_int(), C.c_int()
    w = (C.c_float*4)(*np.asarray(wall, np.float32).reshape(4))
    _lib.lib().ms_host_wall_arc(np.float32(x0), np.float32(y0), np.float32(x0) + np.float32(cell + 2*slack),
                                np.float32(y0) + np.float32(cell + 2*slack), w, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def grid_of(walls, cell=CELL):
    """Origin and dims as cuda.Scenery._build_wall_grid lays them out."""
    fin = walls[np.isfinite(walls).all((1, 2))]
    lo, hi = fin.reshape(-1, 2).min(0), fin.reshape(-1, 2).max(0)
    origin = np.floor(lo) - .5
    dims = np.maximum(np.ceil((hi + .5 - origin)/cell), 1).astype(int)
    return origin.astype(np.float32), dims


def worlds(oracle, walls, poses, keep=None, fov=130., res=64):
    """The oracle's render of one single-agent env per pose (x, y, angle): every env holds the agent's model and the
    walls `keep[i]` selects for pose i (all of them without). Returns the render and, per env, the original wall
    behind each of its static lines."""
    model = scene.agent_model().astype(np.float32)
    M = len(model)
    lines, widths, origin = [], [], []
    for i in range(len(poses)):
        sel = np.arange(len(walls)) if keep is None else np.nonzero(keep[i])[0]
        lines.append(np.concatenate([model, walls[sel]]))
        widths.append(M + len(sel))
        origin.append(sel)
    n_lines = sum(widths)
    sc = oracle.Scene(dict(
        n_agents=1, model=model, lights_vals=np.zeros((0, 3), np.float32), lights_widths=np.zeros(len(poses), np.int32),
        lines_vals=np.concatenate(lines).astype(np.float32), lines_widths=np.array(widths, np.int32),
        textures_vals=np.full((n_lines, 3), .5, np.float32), textures_widths=np.ones(n_lines, np.int32)))
    poses = np.asarray(poses, np.float32)
    agents = dict(angles=poses[:, None, 2], positions=poses[:, None, :2], angvelocity=np.zeros((len(poses), 1), np.float32),
                  velocity=np.zeros((len(poses), 1, 2), np.float32))
    r = oracle.render(sc, agents, oracle.config(core.AGENT_RADIUS, res, fov, 10.))
    # hit indices in terms of the ORIGINAL walls: agent lines stay 0..M-1, wall k of the env becomes M + its number
    idx = r['indices'][:, 0]
    orig = np.full_like(idx, -1)
    for i in range(len(poses)):
        table = np.concatenate([np.arange(M), M + origin[i]])
        orig[i] = np.where(idx[i] >= 0, table[np.maximum(idx[i], 0)], -1)
    return orig, r


def mutated(walls, rng):
    """Walls moved onto each other, collapsed to points, poisoned - the cases the hysteresis rule is sensitive to."""
    w = walls.copy()
    n = len(w)
    pick = rng.choice(n, max(n//8, 4), replace=False)
    w[pick[0::4]] = w[rng.choice(n, len(pick[0::4]))]                  # exact duplicates (coincident walls)
    w[pick[1::4], 1] = w[pick[1::4], 0]                                # zero length
    w[pick[2::4]] += rng.normal(0, 2e-5, w[pick[2::4]].shape).astype(np.float32)   # nearly coincident
    w[pick[3]] = np.nan
    return w


CASES = ['plan0', 'plan1', 'plan2_mutated', 'plan3_mutated', 'plan4', 'plan5_mutated', 'plan6', 'plan7', 'plan8_mutated', 'plan9',
         'large', 'box', 'column', 'oblique0', 'oblique1', 'oblique2_mutated', 'oblique3', 'obliquelarge']


def case_geometry(name):
    """The floorplan behind a case (round 6: `oblique*` - plans turned by seeded angles, with diagonal partitions: the reference's
    walls are exteriors of arbitrary polygons, geometry.py:43-57, and until then every plan-scale case here was axis-aligned)."""
    if name == 'obliquelarge':
        return cubicasa.sample(1, n_unique=16, large=True, oblique=True)[0]
    if name.startswith('oblique'):
        return cubicasa.sample(4, n_unique=16, oblique=True)[int(name[7])]
    if name == 'large':
        return cubicasa.sample(1, n_unique=16, large=True)[0]
    if name.startswith('plan'):
        return cubicasa.sample(10, n_unique=16)[int(name[4])]
    return toys.box() if name == 'box' else toys.column()


def case_walls(name):
    rng = np.random.RandomState(zlib.crc32(name.encode()))              # (not hash(): that changes from run to run)
    if name.startswith('oblique'):
        w = case_geometry(name).walls.astype(np.float32)
        return mutated(w, rng) if name.endswith('mutated') else w
    if name.startswith('plan'):
        g = cubicasa.sample(10, n_unique=16)[int(name[4])]
        w = g.walls.astype(np.float32)
        return mutated(w, rng) if name.endswith('mutated') else w
    if name == 'large':
        return cubicasa.sample(1, n_unique=16, large=True)[0].walls.astype(np.float32)
    return (toys.box() if name == 'box' else toys.column()).walls.astype(np.float32)


@pytest.mark.parametrize('name', CASES)
def test_vis_lists_leave_the_fold_where_it_was(oracle, name):
    walls = case_walls(name)
    rng = np.random.RandomState(1)
    origin, dims = grid_of(walls)
    n_cells = 10 if name.endswith('large') else 24
    cells = rng.choice(dims[0]*dims[1], n_cells, replace=False)
    poses, keep = [], []
    listed = []
    for c in cells:
        vis, _ = scan_cell(walls, origin, dims, c)
        listed.append(vis.mean())
        x0, y0 = origin[0] + (c % dims[0])*CELL, origin[1] + (c//dims[0])*CELL
        for k in range(6):
            # all over the cell, its very edges included
            u = rng.uniform(0, 1, 2) if k < 4 else rng.choice([0., 1.], 2)
            poses.append((x0 + u[0]*CELL, y0 + u[1]*CELL, rng.uniform(-180, 180)))
            keep.append(vis)
    for fov in (130., 164.):
        full, rf = worlds(oracle, walls, poses, fov=fov)
        part, rp = worlds(oracle, walls, poses, keep, fov=fov)
        np.testing.assert_array_equal(part, full, err_msg=f'{name}: hit lines differ at fov {fov}')
        for k in ('distances', 'locations', 'dots'):
            np.testing.assert_array_equal(rp[k], rf[k], err_msg=f'{name}: {k} differ at fov {fov}')
    if name.startswith('plan') or name.startswith('oblique') or name == 'large':
        assert np.mean(listed) < .6, f'the lists hold {np.mean(listed):.2f} of the walls: nothing is being culled'


@pytest.mark.parametrize('name', ['plan0', 'plan2_mutated', 'box', 'oblique0', 'oblique2_mutated'])
def test_near_lists_hold_every_wall_within_reach(name):
    walls = case_walls(name)
    rng = np.random.RandomState(2)
    origin, dims = grid_of(walls)
    a, b = walls[:, 0].astype(np.float64), walls[:, 1].astype(np.float64)
    for c in rng.choice(dims[0]*dims[1], 40, replace=False):
        _, close = scan_cell(walls, origin, dims, c)
        x0, y0 = origin[0] + (c % dims[0])*CELL, origin[1] + (c//dims[0])*CELL
        for u in np.concatenate([rng.uniform(0, 1, (8, 2)), [[0, 0], [1, 0], [0, 1], [1, 1]]]):
            p = np.array([x0 + u[0]*CELL, y0 + u[1]*CELL])
            v = b - a
            with np.errstate(all='ignore'):
                t = np.clip(((p - a)*v).sum(1)/(v*v).sum(1), 0, 1)
            t = np.where(np.isfinite(t), t, 0.)
            d = np.linalg.norm(a + t[:, None]*v - p, axis=1)
            for reach, level in ((REACH, 1), (REACH_LO, 2)):
                must = ~(d > reach)                                    # NaN walls: the reference stops agents at them
                assert not (must & (close < level)).any(), f'{name}: cell {c} misses walls within {reach} m'
        assert (close > 0).sum() < len(walls) or len(walls) < 20


def test_one_wall_hides_another_only_when_it_really_does():
    hidden = lambda cell, o, w, near=NEAR: bool(_lib.lib().ms_host_wall_hidden(
        *map(float, cell), (C.c_float*4)(*o), (C.c_float*4)(*w), near))
    cell = (0., 0., .27, .27)
    wide, target = (2., -3., 2., 3.), (4., -.5, 4., .5)
    assert hidden(cell, wide, target)
    assert hidden(cell, wide[2:] + wide[:2], target)                    # either orientation
    assert not hidden(cell, (2., -3., 2., .2), target)                  # a gap the target shows through
    assert not hidden(cell, (2., -.1, 2., .1), target)                  # too short to try
    assert not hidden(cell, target, wide)                               # the other way round
    assert not hidden(cell, wide, (1., -.5, 1., .5))                    # in front of the occluder
    assert not hidden(cell, wide, (2.001, -.5, 2.001, .5))              # behind it, but inside the hysteresis margin
    assert not hidden((1.8, 0., 2.07, .27), wide, target)               # the cell reaches the occluder's near side
    assert not hidden((1.75, 0., 1.9, .27), wide, target)               # ... or sits inside the near plane
    assert not hidden(cell, wide, (4., .135, 9., .135))                 # the cell straddles the target's own line
    nan = float('nan')
    assert not hidden(cell, (2., nan, 2., 3.), target) and not hidden(cell, wide, (4., nan, 4., .5))
    assert not hidden(cell, wide, (4., -.5, 4., -.5 + 1e-30))


@pytest.mark.parametrize('name', ['plan0', 'plan2_mutated', 'large', 'plan7', 'oblique0', 'oblique2_mutated', 'obliquelarge'])
def test_an_occluder_spans_the_direction_of_what_it_hides(name):
    """The scan's sort of a cell's occluders into sectors of directions (wallgrid_scan_kernel, WG_SECTORS, round 6): a target wall
    is only tried against the occluders of the ONE sector its middle lies in, seen from the cell's centre. That is exact if every
    occluder that hides a wall from the cell (wg_hides: every segment from the cell to the wall crosses it) has that sector in its
    own run - checked here for every (occluder, wall) pair of sampled cells of the case floorplans, cells of both levels of the
    build, the library's own predicates on both sides; and the sort is worth having: a wall's sector holds a fraction of the
    occluders."""
    walls = case_walls(name)
    flat = np.ascontiguousarray(walls.reshape(-1, 4), np.float32)
    rng = np.random.RandomState(9)
    lib = _lib.lib()
    hides = tried = in_sector = 0
    for cell in (CELL, 4*CELL):
        origin, dims = grid_of(walls, cell)
        for c in rng.choice(dims[0]*dims[1], min(6 if name.endswith('large') else 14, dims[0]*dims[1]), replace=False):
            x0, y0 = float(origin[0] + (c % dims[0])*cell - .01), float(origin[1] + (c//dims[0])*cell - .01)
            x1, y1 = x0 + cell + .02, y0 + cell + .02
            pick = rng.choice(len(flat), min(len(flat), 120), replace=False)
            for o in pick:
                oc = (C.c_float*4)(*flat[o])
                for w in pick:
                    if w == o:
                        continue
                    wc = (C.c_float*4)(*flat[w])
                    first, count, sector = C.c_int(), C.c_int(), C.c_int()
                    lib.ms_host_wall_sectors(x0, y0, x1, y1, oc, C.byref(first), C.byref(count), wc, C.byref(sector))
                    inside = (sector.value - first.value) % 64 < count.value
                    tried += 1
                    in_sector += inside
                    if lib.ms_host_wall_hidden(x0, y0, x1, y1, oc, wc, NEAR):
                        hides += 1
                        assert inside, (name, cell, int(c), flat[o], flat[w], first.value, count.value, sector.value)
    assert hides > 500, hides
    assert in_sector < .35*tried, f'{in_sector/tried:.2f} of the (occluder, wall) pairs share a sector: the sort buys nothing'


@pytest.mark.parametrize('name', ['plan0', 'plan2_mutated', 'large', 'box', 'oblique1', 'oblique2_mutated', 'obliquelarge'])
def test_a_wall_is_only_ever_seen_inside_its_arc(name):
    """The view arcs the vis entries carry (wg_arc) and the test the render kernel makes with them (wg_wedge,
    wg_arcs_meet): whenever a ray from a point of the cell can hit a wall at all - its direction is that of some point of
    the wall - a wave whose fan holds that ray keeps the wall, however narrow the fan; and most walls are dropped by a
    narrow fan looking elsewhere."""
    walls = case_walls(name)
    rng = np.random.RandomState(4)
    origin, dims = grid_of(walls)
    meets = _lib.lib().ms_host_wedge_meets
    kept = []
    for c in rng.choice(dims[0]*dims[1], 12, replace=False):
        x0, y0 = origin[0] + (c % dims[0])*CELL, origin[1] + (c//dims[0])*CELL
        for t in rng.choice(len(walls), min(len(walls), 40), replace=False):
            lo, hi = wall_arc(walls[t], origin, dims, c)
            if not np.isfinite(walls[t]).all():
                assert (lo, hi) == (0, 255)
                continue
            for _ in range(12):
                p = np.array([x0, y0]) + rng.choice([0., 1., rng.uniform()], 2)*CELL      # corners and edges too
                q = walls[t, 0] + rng.choice([0., 1., rng.uniform()])*(walls[t, 1] - walls[t, 0])
                d = (q - p).astype(np.float64)
                if not np.abs(d).sum() > 1e-6:
                    continue
                # fans around that direction: a single ray, a sliver, a third of a turn
                for half in (0., 1e-3, .3, 1.):
                    rot = lambda v, a: np.array([v[0]*np.cos(a) - v[1]*np.sin(a), v[0]*np.sin(a) + v[1]*np.cos(a)])
                    off = rng.uniform(-half, half)
                    right, left = rot(d, off - half), rot(d, off + half)
                    assert meets(*np.float32(right), *np.float32(left), lo, hi), (name, c, t)
            # a 30-degree fan looking the other way
            away = -(walls[t].mean(0) - np.array([x0 + CELL/2, y0 + CELL/2]))
            if np.abs(away).sum() > 1.:
                rot = lambda v, a: np.array([v[0]*np.cos(a) - v[1]*np.sin(a), v[0]*np.sin(a) + v[1]*np.cos(a)])
                kept.append(meets(*np.float32(rot(away, -.26)), *np.float32(rot(away, .26)), lo, hi))
    if name != 'box':
        assert np.mean(kept) < .2, f'{np.mean(kept):.2f} of the walls behind a fan are kept'


def test_agents_the_cull_calls_apart_cannot_collide(oracle):
    """ms_physics skips the agent-agent test for pairs its reach cull calls apart: for every such pair the
    oracle's collision_cc (kernels.cu:119-133) must leave x = 1 - at every distance around the threshold, relative
    velocities from a sprint down to 1e-12 a step (project()'s `+ 1e-6` stretches the reach of a pair that moves almost in
    step), pairs in perfect step, coordinates far from the origin, and a NaN or an infinity here and there (apart only when in step)."""
    rng = np.random.RandomState(0)
    lib, cc = _lib.lib(), oracle.lib().oracle_collision_cc
    f32p = C.POINTER(C.c_float)
    n, culled, hits, hits_kept = 60000, 0, 0, 0
    for i in range(n):
        R = float(np.float32(10.**rng.uniform(-2, 0)))
        base = rng.uniform(-1, 1, 2)*10.**rng.uniform(0, 3)
        common = rng.uniform(-1, 1, 2)*10.**rng.uniform(-3, 1)*(rng.rand() < .8)
        rel = 10.**rng.uniform(-12, .7)
        a = rng.uniform(0, 2*np.pi)
        dv = rel*np.array([np.cos(a), np.sin(a)])*(rng.rand() < .95)             # 5 %: in perfect step
        # distances around what the pair can cover: |dv| + 2R, scaled by 0.2 .. 30; now and then much closer / farther
        D = (rel + 2*R)*10.**rng.uniform(-.7, 1.5) if rng.rand() < .8 else 10.**rng.uniform(-3, 2)
        b = a + rng.normal(0, .5) if rng.rand() < .7 else rng.uniform(0, 2*np.pi)   # mostly closing in on each other
        me = np.array([*base, *(common + dv)], np.float32)
        other = np.array([*(base + D*np.array([np.cos(b), np.sin(b)])), *common], np.float32)
        if i % 997 == 0: me[rng.randint(4)] = [np.nan, np.inf, -np.inf][i % 3]
        if i % 1009 == 0: other[rng.randint(4)] = [np.nan, np.inf, -np.inf][i % 3]
        apart = lib.ms_host_agents_apart(me.ctypes.data_as(f32p), other.ctypes.data_as(f32p), R)
        x = cc(*[float(v) for v in me], *[float(v) for v in other], R)
        hits += x < 1
        if apart:
            culled += 1
            assert x == 1., (i, me, other, R, x)
            assert (np.isfinite(me).all() and np.isfinite(other).all()) or (me[2:] == other[2:]).all()   # (in step: never collide)
        else:
            hits_kept += x < 1
    assert hits > n//20 and hits_kept == hits, 'the sample holds plenty of collisions, none of them culled'
    assert culled > n//4, 'and the cull does cull'


def test_walls_the_cull_calls_beyond_reach_cannot_stop_the_agent(oracle):
    """ms_physics skips the agent-wall test for walls its reach cull calls beyond the agent's reach: for
    every such pair the oracle's collision_cs (kernels.cu:135-171) must leave x = 1 - walls at every distance around the
    reach, agents from a sprint down to 1e-12 a step (project()'s `+ 1e-6`: a crawling agent is stopped by walls metres
    away), walls from 10 m down to a micrometre (its `+ 1e-6` on the wall's length), a NaN or an infinity now and then."""
    rng = np.random.RandomState(1)
    lib, cs = _lib.lib(), oracle.lib().oracle_collision_cs
    f32p = C.POINTER(C.c_float)
    n, culled, hits, hits_kept = 80000, 0, 0, 0
    for i in range(n):
        R = float(np.float32(10.**rng.uniform(-2, 0)))
        p = rng.uniform(-1, 1, 2)*10.**rng.uniform(0, 3)
        speed = 10.**rng.uniform(-12, .7)*(rng.rand() < .97)                     # 3 %: standing still
        a = rng.uniform(0, 2*np.pi)
        v = speed*np.array([np.cos(a), np.sin(a)])
        reach = (speed + 2*R)*(1 + 1e-6/max(speed, 1e-30))**2 if speed > 0 else 2*R
        D = min(reach, 1e4)*10.**rng.uniform(-.7, 1.) if rng.rand() < .8 else 10.**rng.uniform(-3, 2)
        b = a + rng.normal(0, .6) if rng.rand() < .7 else rng.uniform(0, 2*np.pi)   # mostly ahead of the agent
        mid = p + D*np.array([np.cos(b), np.sin(b)])
        length = 10.**rng.uniform(-6, 1)
        c = rng.uniform(0, 2*np.pi)
        half = .5*length*np.array([np.cos(c), np.sin(c)])
        shift = rng.uniform(-1, 1)*half*(rng.rand() < .5)                         # the nearest point: an end as often as not
        agent = np.array([*p, *v], np.float32)
        wall = np.array([*(mid - half + shift), *(mid + half + shift)], np.float32)
        if i % 997 == 0: agent[rng.randint(4)] = [np.nan, np.inf, -np.inf][i % 3]
        if i % 1009 == 0: wall[rng.randint(4)] = [np.nan, np.inf, -np.inf][i % 3]
        beyond = lib.ms_host_wall_beyond_reach(agent.ctypes.data_as(f32p), wall.ctypes.data_as(f32p), R)
        x = cs(*[float(t) for t in agent], *[float(t) for t in wall], R)
        hits += x < 1
        if beyond:
            culled += 1
            assert x == 1., (i, agent, wall, R, x)
            assert np.isfinite(agent[:2]).all() and np.isfinite(wall).all()      # (a NaN velocity stops nobody: every test of collision_cs fails)
        else:
            hits_kept += x < 1
    assert hits > n//20 and hits_kept == hits, 'the sample holds plenty of collisions, none of them culled'
    assert culled > n//4, 'and the cull does cull'


def _groups_for(res):
    """How many 64-ray groups a wave may be given at this resolution (render_kernel's NG; ms_debug_ray_groups)."""
    return 4 if res > 128 else 2 if res > 64 else 1


@pytest.mark.parametrize('res,fov', [(64, 130.), (512, 130.), (200, 60.), (7, 170.), (1, 90.), (128, 165.)])
def test_no_ray_outside_a_lines_interval_hits_it(oracle, res, fov):
    """render_kernel's pass 1 turns a line into an interval of the wave's rays and intersects it with no others
    (DESIGN.md 3.2): every ray the oracle's raycast (kernels.cu:352-377, which tries them all) lands on the line must be
    inside - lines near and far, behind the agent, through its near plane, across the edges of the fan, end-on, tiny."""
    from megastep_amd import scene as scene_
    rng = np.random.RandomState(res)
    lib = _lib.lib()
    E, M = 3000, len(scene_.agent_model())
    radius = float(np.float32(.15/2**.5))
    pos = rng.uniform(-5, 5, (E, 2)).astype(np.float32)
    ang = rng.uniform(-180, 180, E).astype(np.float32)
    # a wall somewhere around the agent: its middle at a log-uniform distance in a direction anywhere, any orientation
    dist = 10.**rng.uniform(-1.3, 1.3, E)
    where = rng.uniform(0, 2*np.pi, E)
    mid = pos + (dist*np.array([np.cos(where), np.sin(where)])).T
    length = 10.**rng.uniform(-3, 1.3, E)
    along = np.where(rng.rand(E) < .3, where + rng.normal(0, .02, E), rng.uniform(0, 2*np.pi, E))   # a third: end-on
    half = (.5*length*np.array([np.cos(along), np.sin(along)])).T
    walls = np.concatenate([mid - half, mid + half], 1).astype(np.float32).reshape(E, 1, 2, 2)
    model = scene_.agent_model()
    lines = np.concatenate([np.tile(model[None], (E, 1, 1, 1)), walls], 1)              # (agent rows are redrawn by the oracle)
    texw = np.full(E*(M + 1), 4, np.int32)
    S = oracle.Scene(dict(n_agents=1, model=model, lights_vals=np.zeros((0, 3)), lights_widths=[0]*E,
                          lines_vals=lines.reshape(-1, 2, 2), lines_widths=[M + 1]*E,
                          textures_vals=np.full((texw.sum(), 3), .5), textures_widths=texw))
    ag = dict(angles=ang.reshape(E, 1), positions=pos.reshape(E, 1, 2), angvelocity=np.zeros((E, 1), np.float32),
              velocity=np.zeros((E, 1, 2), np.float32))
    idx = oracle.render(S, ag, oracle.config(radius, res, fov, 10))['indices'][:, 0]
    f32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int)
    hits = covered = 0
    for e in range(E):
        s_, c_ = C.c_float(), C.c_float()
        lib.ms_host_sincospi(float(np.float32(ang[e])/np.float32(180.)), C.byref(s_), C.byref(c_))
        pose = np.array([pos[e, 0], pos[e, 1], s_.value, c_.value], np.float32)
        line = walls[e].reshape(4)
        # a wave serves 64 rays, or - as ms_render picks above 64 rays - 128 or 256 of them (render_kernel's NG)
        for groups in sorted({1, _groups_for(res)}):
            W = 64*groups
            for g in range((res + W - 1)//W):
                lo, n = C.c_int(), C.c_int()
                lib.ms_host_ray_interval_wide(pose.ctypes.data_as(f32p), line.ctypes.data_as(f32p), res, fov, radius, groups, g, C.byref(lo), C.byref(n))
                if groups == 1:
                    lo1, n1 = C.c_int(), C.c_int()
                    lib.ms_host_ray_interval(pose.ctypes.data_as(f32p), line.ctypes.data_as(f32p), res, fov, radius, g, C.byref(lo1), C.byref(n1))
                    assert (lo1.value, n1.value) == (lo.value, n.value)
                rays = np.nonzero(idx[e, W*g:W*g + W] == M)[0]
                assert n.value == 0 or (lo.value >= 0 and lo.value + n.value <= min(W, res - W*g)), (e, g, lo.value, n.value)
                assert len(rays) == 0 or (rays.min() >= lo.value and rays.max() < lo.value + n.value), (e, groups, g, rays, lo.value, n.value, pose, line)
                if groups == 1:
                    hits += len(rays)
                    covered += n.value
    assert hits > E*res//200, 'plenty of rays land on their wall'
    assert covered < 8*hits + E, 'and the intervals are not much wider than what they must hold'


def test_three_key_slots_settle_a_ray_like_the_literal_fold():
    """The reference folds a ray's hits in LINE order with a hysteresis - `if (near < s && s < x - 1e-4) x = s`
    (kernels.cu:369-376) - so its answer depends on that order; render_kernel meets the hits in whatever order its lists
    and its 64 lanes bring them and keeps three keys per ray.  Hits in clusters inside the band,
    coincident walls, duplicates of one distance, far-apart hits; every set played in several random orders, lockstep
    window by window: either the slots say they cannot tell (the kernel then folds literally) or they give the fold's hit."""
    rng = np.random.RandomState(0)
    lib = _lib.lib()
    f32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int)
    settled = unsure = 0
    for trial in range(6000):
        n = int(rng.choice([1, 2, 3, 4, 6, 10, 40, 100, 200]))
        kind = trial % 4
        if kind == 0:                                  # scattered
            s = rng.uniform(.2, 20., n)
        elif kind == 1:                                # a cluster in the band at the front, others behind
            s = np.concatenate([3. + rng.uniform(0, 3e-4, (n + 1)//2), rng.uniform(3., 9., n//2)])
        elif kind == 2:                                # steps just under / just over the hysteresis, a chain of them
            s = 5. - np.cumsum(rng.choice([.9e-4, .99e-4, 1.01e-4, 1.1e-4, 2e-4], n))
        else:                                          # exact duplicates and near-duplicates
            s = rng.choice(np.float32([2., 2. + 1e-4, 2. + 5e-5, 2. - 1e-4, 7.]), n) + rng.choice([0., 0., 1e-7], n)
        s = np.ascontiguousarray(s, np.float32)
        line = np.ascontiguousarray(rng.permutation(4*n)[:n], np.int32)          # distinct lines, any order
        # the literal fold, in line order (float32 arithmetic, as the kernels')
        x, xi = np.float32(np.inf), -1
        for k in np.argsort(line):
            if s[k] < x - np.float32(1e-4):
                x, xi = s[k], int(line[k])
        for rep in range(4):
            order = np.ascontiguousarray(rng.permutation(n), np.int32)
            out_s, out_i = C.c_float(), C.c_int()
            amb = lib.ms_host_fold_hits(s.ctypes.data_as(f32p), line.ctypes.data_as(i32p), n, order.ctypes.data_as(i32p),
                                        C.byref(out_s), C.byref(out_i))
            if amb:
                unsure += 1
            else:
                settled += 1
                assert out_i.value == xi and np.float32(out_s.value) == x, (trial, rep, s, line, order, out_s.value, out_i.value, x, xi)
    assert settled > unsure > 1000, 'settled by the slots more often than not, even here; and the sample does reach the cases they leave to the fold'


@pytest.mark.parametrize('name', ['plan0', 'plan1', 'plan3', 'large', 'box', 'column', 'oblique0', 'oblique3', 'obliquelarge'])
def test_light_grid_verdicts_and_candidates_hold_on_every_sampled_point(name):
    """The light grid (ms_bake) may call a light LIT or DARK for a cell only where the reference's
    obstructed() (kernels.cu:253-257) says so at EVERY point of the cell, and the candidate walls it lists for the lights
    it leaves open must reproduce obstructed()'s verdict over all walls at every point - checked on the library's host
    instantiation of the build (the predicates the gfx950 kernels are compiled from), points all over the cell and on its
    corners.  (tests/test_gpu_parity.py does the same with the grid a GPU built.)"""
    from tests import util
    rng = np.random.RandomState(3)
    if name.startswith('plan'):
        g = cubicasa.sample(4, n_unique=16)[int(name[4])]
    else:
        g = case_geometry(name)
    walls = np.ascontiguousarray(g.walls, np.float32)
    pos = np.asarray(g.lights, np.float32).reshape(-1, 2)
    lo, hi = walls.reshape(-1, 2).min(0), walls.reshape(-1, 2).max(0)
    extra = rng.uniform(lo, hi, (6, 2)).astype(np.float32)                    # a few lights anywhere, inside walls or not
    lights = np.ascontiguousarray(np.concatenate([np.concatenate([pos, extra])[:64], np.ones((min(len(pos) + 6, 64), 1), np.float32)], 1), np.float32)
    origin, dims = grid_of(walls)
    lib = _lib.lib()
    flat = np.ascontiguousarray(walls.reshape(-1, 4))
    n_lit = n_dark = n_open = 0
    for c in rng.choice(dims[0]*dims[1], min({'large': 40, 'obliquelarge': 100}.get(name, 120), dims[0]*dims[1]), replace=False):
        words, cands = np.zeros(4, np.uint32), np.zeros(4096, np.uint32)
        n = lib.ms_host_lightgrid_cell(flat.ctypes.data, len(flat), lights.ctypes.data, len(lights), float(origin[0]), float(origin[1]),
                                       int(dims[0]), int(dims[1]), CELL, int(c), words.ctypes.data, cands.ctypes.data, len(cands))
        assert 0 <= n <= len(cands)
        cands = cands[:n]
        assert ((cands >> 31) == 1).all()
        x0, y0 = origin[0] + (c % dims[0])*CELL, origin[1] + (c//dims[0])*CELL
        pts = (np.array([x0, y0]) + rng.uniform(0, CELL, (24, 2))).astype(np.float32)
        pts = np.concatenate([pts, np.float32([[x0, y0], [x0 + CELL, y0], [x0, y0 + CELL], [x0 + CELL, y0 + CELL]])])
        for i, light in enumerate(lights):
            state = (int(words[i >> 4]) >> (2*(i & 15))) & 3
            blocked = util.obstructed(light[:2], pts, walls).any(1)
            if state == 1:
                assert not blocked.any(), (name, c, i, 'LIT cell has a shadowed point')
                n_lit += 1
            elif state == 2:
                assert blocked.all(), (name, c, i, 'DARK cell has a lit point')
                n_dark += 1
            else:
                mine = [int(k & 0xffffff) for k in cands if (int(k) >> 24) & 63 == i]
                few = util.obstructed(light[:2], pts, walls[mine]).any(1) if mine else np.zeros(len(pts), bool)
                assert (few == blocked).all(), (name, c, i, 'candidate list misses a blocker')
                n_open += 1
    assert n_lit > 20 and n_dark > 20 and n_open > 20, (n_lit, n_dark, n_open)


@pytest.mark.parametrize('name,res,fov', [('plan0', 64, 130.), ('plan1', 256, 130.), ('plan2_mutated', 128, 90.), ('plan6', 512, 60.),
                                          ('large', 64, 160.), ('column', 32, 130.), ('oblique0', 64, 130.), ('oblique1', 256, 130.),
                                          ('oblique2_mutated', 128, 70.), ('obliquelarge', 256, 130.)])
def test_the_whole_chain_of_culls_keeps_every_winning_pair(oracle, name, res, fov):
    """What a render wave intersects, put together on the host from the pieces the kernel is compiled from: the walls on
    the vis list of the agent's cell, less those whose view arc misses the run of directions of the wave's own rays
    (first ray to last), each with its interval of the wave's rays.  For poses all over sampled cells the oracle's raycast
    of the WHOLE world (kernels.cu:352-377) must land every ray on a (wall, ray) pair of that set - and, since the set
    only ever leaves out pairs that do not intersect, the fold over it then ends where the fold over everything does."""
    walls = case_walls(name)
    rng = np.random.RandomState(7)
    lib = _lib.lib()
    origin, dims = grid_of(walls)
    M = len(scene.agent_model())
    cells = rng.choice(dims[0]*dims[1], 6 if name.endswith('large') else 16, replace=False)
    poses, cell_of = [], []
    for c in cells:
        x0, y0 = origin[0] + (c % dims[0])*CELL, origin[1] + (c//dims[0])*CELL
        for k in range(4):
            u = rng.uniform(0, 1, 2) if k < 3 else rng.choice([0., 1.], 2)
            poses.append((x0 + u[0]*CELL, y0 + u[1]*CELL, rng.uniform(-180, 180)))
            cell_of.append(c)
    full, _ = worlds(oracle, walls, poses, fov=fov, res=res)
    f32, f32p, i32p = np.float32, C.POINTER(C.c_float), C.POINTER(C.c_int)
    half_screen = f32(np.tan(np.pi/180*fov/2))
    lists = {c: (scan_cell(walls, origin, dims, c)[0], [wall_arc(w, origin, dims, c) for w in walls]) for c in cells}
    kept_pairs = all_pairs = on_walls = 0
    for i, (x, y, a) in enumerate(poses):
        vis, arcs = lists[cell_of[i]]
        s_, c_ = C.c_float(), C.c_float()
        lib.ms_host_sincospi(float(f32(a)/f32(180.)), C.byref(s_), C.byref(c_))
        sn, cs = f32(s_.value), f32(c_.value)
        pose = np.array([x, y, sn, cs], f32)
        r = np.arange(res, dtype=f32)
        uy = (f32(res) - f32(2)*r - f32(1))*half_screen/f32(res)           # ray_y, kernels.cu:234-236, in float32 as the kernel
        rx, ry = cs*f32(1) - sn*uy, sn*f32(1) + cs*uy
        W = 64*_groups_for(res)                                              # the rays of one wave, as ms_render deals them
        for g in range((res + W - 1)//W):
            first, last = W*g, min(W*g + W - 1, res - 1)
            allowed = np.zeros((len(walls), last - first + 1), bool)
            for t in np.nonzero(vis)[0]:
                lo8, hi8 = arcs[t]
                # rays run from the left of the view (ray 0) to its right: the wave's rightmost ray is its last
                if not lib.ms_host_wedge_meets(float(rx[last]), float(ry[last]), float(rx[first]), float(ry[first]), lo8, hi8):
                    continue
                lo, n = C.c_int(), C.c_int()
                lib.ms_host_ray_interval_wide(pose.ctypes.data_as(f32p), np.ascontiguousarray(walls[t].reshape(4)).ctypes.data_as(f32p),
                                              res, fov, float(core.AGENT_RADIUS), W//64, g, C.byref(lo), C.byref(n))
                allowed[t, lo.value:lo.value + n.value] = True
            hit = full[i, first:last + 1]
            for k in np.nonzero(hit >= M)[0]:
                assert allowed[hit[k] - M, k], (name, i, g, k, int(hit[k] - M), poses[i])
            on_walls += int((hit >= M).sum())
            kept_pairs += int(allowed.sum())
            all_pairs += allowed.size
    assert on_walls > (50 if name == 'column' else len(poses)*res//3), 'rays do land on walls'
    assert kept_pairs < .2*all_pairs, f'{kept_pairs/all_pairs:.2f} of all (wall, ray) pairs are kept: nothing is being culled'


@pytest.mark.parametrize('name', ['plan0', 'plan2_mutated', 'plan7', 'box', 'oblique1', 'oblique2_mutated'])
def test_the_walls_a_physics_wave_meets_are_all_that_can_stop_the_agent(oracle, name):
    """ms_physics' chain put together on the host: the agent's reach picks the short or the long tier of its cell's near
    list (or, past the long one, every wall), the reach cull drops what is beyond, the rest goes through the exact test -
    and the least progress over those must be the least over ALL the env's walls (kernels.cu:202-221), for agents
    anywhere in sampled cells, from a sprint that outruns the lists down to a crawl whose reach outgrows them."""
    walls = case_walls(name)
    rng = np.random.RandomState(9)
    lib, cs = _lib.lib(), oracle.lib().oracle_collision_cs
    f32p = C.POINTER(C.c_float)
    origin, dims = grid_of(walls)
    R = float(core.AGENT_RADIUS)
    flat = [np.ascontiguousarray(w.reshape(4), np.float32) for w in walls]
    listed = swept = stops = met = 0
    for c in rng.choice(dims[0]*dims[1], 30, replace=False):
        _, close = scan_cell(walls, origin, dims, c)
        x0, y0 = origin[0] + (c % dims[0])*CELL, origin[1] + (c//dims[0])*CELL
        for k in range(8):
            u = rng.uniform(0, 1, 2) if k < 6 else rng.choice([0., 1.], 2)
            speed = 10.**rng.uniform(-9, .3)                                    # per step: a nanometre to two metres
            a = rng.uniform(0, 2*np.pi)
            agent = np.array([x0 + u[0]*CELL, y0 + u[1]*CELL, speed*np.cos(a), speed*np.sin(a)], np.float32)
            ap = agent.ctypes.data_as(f32p)
            reach = lib.ms_host_wall_reach(ap, R)
            if reach <= REACH_LO: mine = np.nonzero(close >= 2)[0]
            elif reach <= REACH: mine = np.nonzero(close >= 1)[0]
            else: mine = np.arange(len(walls))                                  # the sweep
            listed += reach <= REACH; swept += reach > REACH
            x_all = min((cs(*[float(t) for t in agent], *[float(t) for t in w], R) for w in flat), default=1.)
            x_mine = 1.
            for t in mine:
                if not lib.ms_host_wall_beyond_reach(ap, flat[t].ctypes.data_as(f32p), R):
                    met += reach <= REACH
                    x_mine = min(x_mine, cs(*[float(v) for v in agent], *[float(v) for v in flat[t]], R))
            assert x_mine == x_all, (name, c, k, agent, reach, x_mine, x_all)
            stops += x_all < 1
    assert listed > 100 and swept > 5 and stops > 20, (listed, swept, stops)
    if name != 'box':
        assert met < .1*listed*len(walls), 'an agent whose reach the lists cover meets a fraction of the walls'
