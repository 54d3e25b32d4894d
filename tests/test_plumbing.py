"""BASELINE config 1: 1 env x 1 agent, single-room geometry - scene/geometry build + Ragged pack on CPU (no GPU)."""
import numpy as np
import pytest
import torch
from megastep_amd import core, cuda, cubicasa, geometry, ragged, scene, sharding, spaces, toys


def test_box_room_scenery_on_cpu():
    g = toys.box()
    assert g.walls.shape == (4, 2, 2) and g.masks.shape == (36, 36) and g.masks.dtype == np.int16 and g.res == .2
    np.testing.assert_allclose(g.walls, [[[6, 6], [1, 6]], [[1, 6], [1, 1]], [[1, 1], [6, 1]], [[6, 1], [6, 6]]], atol=1e-12)
    assert set(np.unique(g.masks)) == {-1, 0, 1}
    inside = geometry.centers(np.stack((g.masks == 1).nonzero(), -1), g.masks.shape, g.res)
    assert (inside > 1).all() and (inside < 6).all()
    s = scene.scenery([g], n_agents=1, device='cpu', bake=False)
    assert isinstance(s.lines, cuda.Ragged3D) and isinstance(s.lights, cuda.Ragged2D) and isinstance(s.baked, cuda.Ragged1D)
    assert s.lines.vals.shape == (12, 2, 2) and s.lines.widths.tolist() == [12]
    assert s.textures.vals.shape == (416, 3) and s.textures.widths.tolist() == [2]*8 + [100]*4
    assert s.baked.vals.shape == (416,) and (s.baked.vals == 1).all()
    assert s.lights.vals.shape == (1, 3) and .5 <= s.lights.vals[0, 2] <= 2
    assert s.model.shape == (8, 2, 2) and s.n_agents == 1
    for r in (s.lines, s.lights, s.textures):
        assert r.widths.dtype == r.starts.dtype == r.ends.dtype == r.inverse.dtype == torch.int32
    st = s.state(0)
    assert st.lines.shape == (12, 2, 2) and len(st.textures) == 12 and st.baked.vals.shape == (416,)


def test_core_on_cpu_holds_state_but_cannot_step():
    s = scene.scenery(3*[toys.box()], n_agents=2, device='cpu', bake=False)
    c = core.Core(s, res=32, fov=100, fps=20)
    assert (c.n_envs, c.n_agents, c.res, c.fov, c.fps) == (3, 2, 32, 100, 20)
    assert c.agents.positions.shape == (3, 2, 2) and c.progress.shape == (3, 2) and (c.progress == 1).all()
    assert c.env_full(True).dtype == torch.bool and c.agent_full(1.).shape == (3, 2) and c.env_full(3).dtype == torch.int32
    c.agents.positions[:] = torch.tensor([3., 3.])
    assert (c.agents.state(1).positions == 3).all()
    st = c.state(0)
    assert st.n_agents == 2 and st.agents.angles.shape == (2,) and st.scenery.lines.shape == (20, 2, 2)
    with pytest.raises(AssertionError):
        core.Core(s, fov=180)


def test_ragged_validation_and_edges():
    vals, widths = torch.arange(6.), torch.tensor([3, 1, 2], dtype=torch.int32)
    r = ragged.Ragged(vals, widths)
    assert isinstance(r, cuda.Ragged1D) and r[1].tolist() == [3.] and r[-1].tolist() == [4., 5.]
    assert r[:2].vals.tolist() == [0., 1., 2., 3.] and r[1:].widths.tolist() == [1, 2]
    assert r.clone().vals.data_ptr() != r.vals.data_ptr()
    assert isinstance(r.numpyify(), ragged.RaggedNumpy)
    z = ragged.Ragged(torch.arange(5.), torch.tensor([2, 0, 3, 0], dtype=torch.int32))      # zero-width rows
    assert z.starts.tolist() == [0, 2, 2, 5] and z.inverse.tolist() == [0, 0, 2, 2, 2] and z[1].numel() == 0
    with pytest.raises(RuntimeError):
        ragged.Ragged(vals, torch.tensor([3, 1, 1], dtype=torch.int32))
    with pytest.raises(RuntimeError):
        ragged.Ragged(vals, widths.long())
    with pytest.raises(RuntimeError):
        ragged.Ragged(vals.double(), widths)
    with pytest.raises(RuntimeError):
        ragged.Ragged(torch.zeros(6, 2).t()[0:1].t(), torch.tensor([6], dtype=torch.int32)[:0])
    with pytest.raises(RuntimeError):
        ragged.Ragged(torch.zeros(2, 2, 2, 2), torch.tensor([2], dtype=torch.int32))
    assert isinstance(ragged.Ragged(np.arange(6), np.array([3, 1, 2])), ragged.RaggedNumpy)
    assert isinstance(ragged.RaggedNumpy(np.arange(6.), np.array([3, 1, 2])).torchify(), cuda.Ragged1D)


def test_scenery_and_agents_validation():
    s = scene.scenery([toys.box()], 1, device='cpu', bake=False)
    with pytest.raises(RuntimeError):
        cuda.Scenery(1, s.lights, s.lines, s.lights, s.model)                      # textures not per line
    with pytest.raises(RuntimeError):
        cuda.Agents(torch.zeros(2, 1), torch.zeros(2, 1, 3), torch.zeros(2, 1), torch.zeros(2, 1, 2))
    with pytest.raises(RuntimeError):
        cuda.Agents(torch.zeros(2, 1).double(), torch.zeros(2, 1, 2), torch.zeros(2, 1), torch.zeros(2, 1, 2))
    with pytest.raises(RuntimeError):
        cuda.initialize(.1, 64, 190, 10)


def test_every_core_carries_its_own_constants():
    """VERDICT r4 / SURVEY 5 "no globals => multi-device safe": the C-ABI takes MsConfig by value, and the Python side no longer
    funnels it through one process-global either - a Core keeps its own and hangs it on its Agents; initialize()'s global is the
    fallback for bare two-argument calls on hand-built Agents (the reference's only mode, kernels.cu:12-27)."""
    from megastep_amd import core
    s = scene.scenery([toys.box()], 1, device='cpu', bake=False)
    a = core.Core(s, res=64, fov=130, fps=10)
    b = core.Core(s, res=256, fov=70, fps=20)
    assert (a.config.res, a.config.fov, a.config.fps) == (64, 130., 10.) and (b.config.res, b.config.fov) == (256, 70.)
    assert cuda._cfg(a.agents) is a.config and cuda._cfg(b.agents) is b.config          # whichever Core came last
    bare = cuda.Agents(a.agents.angles, a.agents.positions, a.agents.angvelocity, a.agents.velocity)
    assert cuda._cfg(bare).res == 256                                                    # the fallback: the last initialize()
    assert cuda._cfg(bare, a.config) is a.config and cuda._cfg(b.agents, a.config) is a.config   # config= wins
    view = cuda.Agents(a.agents.angles, a.agents.positions, a.agents.angvelocity, a.agents.velocity, config=a.config)
    assert cuda._cfg(view) is a.config
    with pytest.raises(RuntimeError):
        cuda._cfg(bare, (0.1, 64, 130., 10.))
    with pytest.raises(RuntimeError):
        cuda.config(.1, 64, 190, 10)


def test_the_light_grid_stays_inside_its_byte_budget(monkeypatch):
    """at 0.125 m cells with the candidates' rows a cell costs 264 bytes - 13 GB for 4096 plans, 54 GB for large ones -
    and there was no cap. Over LIGHT_GRID_BYTES the grid first drops the candidates' rows (ms_render handles NULL), then doubles
    its cells until it fits; what was built is in grid_report()."""
    g = cubicasa.sample(6, n_unique=16, seed=2)
    s = scene.scenery(g, 2, device='cpu', bake=False)
    full = s._light_grid()
    rep = s.grid_report()['light_grid']
    assert rep['cell'] == cuda.Scenery.LIGHT_GRID_CELL and rep['candidate_rows'] and full[7] is not None and rep['floorplans'] == 6
    assert rep['bytes'] == 24*(rep['cells'] + 1) + 20*full[6].shape[0]                      # verdicts + headers, pool words + their rows
    monkeypatch.setattr(cuda.Scenery, 'LIGHT_GRID_BYTES', rep['bytes'] - 1)
    lean = s._light_grid()
    r = s.grid_report()['light_grid']
    assert lean[7] is None and not r['candidate_rows'] and r['cell'] == rep['cell'] and r['bytes'] < rep['bytes']//3
    monkeypatch.setattr(cuda.Scenery, 'LIGHT_GRID_BYTES', rep['bytes']//20)
    coarse = s._light_grid()
    r = s.grid_report()['light_grid']
    assert r['cell'] > rep['cell'] and r['bytes'] <= rep['bytes']//20 and coarse[3] == r['cell'] and coarse[0].shape[0] == r['cells'] + 1
    # a shard carries the lean grid over like any other (no rows to repack)
    assert s.grid_report()['wall_grid'] is None                                            # (never baked: no wall grid)


def test_column_and_spaces():
    g = toys.column()
    np.testing.assert_allclose(sorted(g.lights.tolist()), [[2.5, 2.5], [2.5, 4.5], [4.5, 2.5], [4.5, 4.5]], atol=1e-12)
    assert np.abs(g.walls - 3.5).max() == pytest.approx(.05)
    assert spaces.MultiImage(2, 3, 1, 64).shape == (2, 3, 1, 64) and spaces.MultiDiscrete(1, 7).shape == (1, 7)
    assert spaces.MultiVector(4, 3).shape == (4, 3) and spaces.MultiConstant(2).shape == (2,)


def test_masks_rasteriser():
    walls = np.array([[[1.1, 1.1], [3.1, 1.1]], [[3.1, 1.1], [3.1, 2.3]]])
    room = np.array([[1.1, 1.1], [3.1, 1.1], [3.1, 2.3], [1.1, 2.3]])
    m = geometry.masks(walls, [room])
    H, W = m.shape
    assert (H, W) == (int(3.3/.2) + 1, int(4.1/.2) + 1)
    cells = lambda x, y: m[geometry.indices(np.array([x, y]), m.shape, .2)[0], geometry.indices(np.array([x, y]), m.shape, .2)[1]]
    assert cells(2., 1.1) == -1 and cells(3.1, 2.) == -1          # on the walls
    assert cells(2., 1.7) == 1 and cells(.5, .5) == 0             # inside the room / outside


def test_synthetic_cubicasa_sample_is_deterministic_and_in_range():
    a, b = cubicasa.sample(6, n_unique=32), cubicasa.sample(6, n_unique=32)
    for ga, gb in zip(a, b):
        assert ga.id == gb.id and (ga.walls == gb.walls).all() and (ga.masks == gb.masks).all()
    for g in a:
        assert set(g.keys()) == {'id', 'walls', 'lights', 'masks', 'res'}
        assert 150 <= len(g.walls) <= 404 and 10 <= len(g.lights) <= 25 and g.walls.min() > geometry.MARGIN - 1e-9
        assert scene.lengths(g.walls).min() > 1e-3
        assert (g.masks > 0).sum() > 100 and g.masks.min() == -1
        room_cells = geometry.indices(g.lights, g.masks.shape, g.res)
        assert (g.masks[room_cells[:, 0], room_cells[:, 1]] > 0).all()       # every light sits inside a room
    assert len({g.id for g in cubicasa.sample(40, n_unique=32)}) <= 29        # repeats cyclically over the split
    assert cubicasa.sample(3, split='test', n_unique=32)[0].id != a[0].id
    with pytest.raises(ValueError):
        cubicasa.sample(1, split='validation')
    big = cubicasa.sample(1, large=True, n_unique=16)[0]
    assert 800 <= len(big.walls) <= 1204


def test_oblique_plans_are_the_aligned_ones_turned_with_diagonal_partitions():
    """cubicasa.sample(oblique=True) (round 6; the reference's walls are exteriors of arbitrary polygons, geometry.py:43-57): the
    plan of the same index, a few diagonal pieces added, everything turned by one seeded angle - walls at two families of
    directions 90 degrees apart plus the diagonals', every coordinate still beyond MARGIN, lights in their rooms, lengths kept."""
    a, b = cubicasa.sample(5, n_unique=32, oblique=True), cubicasa.sample(5, n_unique=32, oblique=True)
    flat = cubicasa.sample(5, n_unique=32)
    for ga, gb, g0 in zip(a, b, flat):
        assert ga.id == gb.id and ga.id != g0.id and (ga.walls == gb.walls).all() and (ga.masks == gb.masks).all()
        extra = len(ga.walls) - len(g0.walls)
        assert extra >= 8 and extra % 4 == 0                                  # the diagonal pieces: four segments each
        assert ga.walls.min() > geometry.MARGIN - 1e-9 and scene.lengths(ga.walls).min() > 1e-3
        # a turn keeps lengths: the aligned plan's walls come first, in their order
        np.testing.assert_allclose(scene.lengths(ga.walls[:len(g0.walls)]), scene.lengths(g0.walls), atol=1e-9)
        d = ga.walls[:, 1] - ga.walls[:, 0]
        ang = np.degrees(np.arctan2(d[:, 1], d[:, 0])) % 90
        main = np.median(ang[:len(g0.walls)])
        assert (np.abs(ang[:len(g0.walls)] - main) < 1e-6).mean() > .99        # one turn for the whole plan
        assert (np.abs(ang[len(g0.walls):] - main) > 1.).any()                 # ... and the diagonals are at others
        assert (np.abs(d).min(1) > 1e-3).mean() > .9, 'hardly anything is left aligned with the axes'
        room_cells = geometry.indices(ga.lights, ga.masks.shape, ga.res)
        assert (ga.masks[room_cells[:, 0], room_cells[:, 1]] > 0).all() and (ga.masks > 0).sum() > 100 and ga.masks.min() == -1
    big = cubicasa.sample(1, large=True, n_unique=16, oblique=True)[0]
    assert 800 <= len(big.walls) <= 1300


def test_env_slices_partition_the_envs():
    for n, w in [(4096, 8), (10, 3), (7, 7), (5, 8)]:
        slices = [sharding.env_slice(n, r, w) for r in range(w)]
        assert slices[0][0] == 0 and slices[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(slices, slices[1:]))
        sizes = [b - a for a, b in slices]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.env_slice(8, 8, 8)


def test_shard_scenery_reassembles():
    gs = cubicasa.sample(5, n_unique=16)
    full = scene.scenery(gs, 2, device='cpu', random=np.random.RandomState(0), bake=False)
    full.baked.vals.copy_(torch.rand_like(full.baked.vals))
    shards = [sharding.shard_scenery(full, r, 2) for r in range(2)]
    assert [len(s.lines) for s in shards] == [3, 2]
    for name in ('lights', 'lines', 'textures', 'baked'):
        vals = torch.cat([getattr(s, name).vals for s in shards])
        widths = torch.cat([getattr(s, name).widths for s in shards])
        assert torch.equal(vals, getattr(full, name).vals) and torch.equal(widths, getattr(full, name).widths)
    assert shards[1].lines.starts[0] == 0 and shards[1].textures.starts[0] == 0


def test_light_grid_shard_repacks_candidate_lists():
    """sharding._shard_light_grid: a shard's cells keep their verdicts and candidate lists; the lists move to a
    pool of the shard's own (the parent's pool is filled in no particular order). Pure tensor logic, so it runs here."""
    import torch
    from megastep_amd import sharding
    rng = np.random.RandomState(0)
    dims = np.array([[3, 2], [1, 4], [2, 2], [5, 1]])
    cells = dims.prod(1)
    starts = torch.tensor(np.concatenate([[0], np.cumsum(cells)[:-1]]), dtype=torch.int32)
    geom = torch.tensor(np.concatenate([rng.uniform(0, 5, (4, 2)), dims], 1), dtype=torch.float32)
    total = int(cells.sum())
    vals = torch.tensor(rng.randint(0, 2**31 - 1, (total, 4)), dtype=torch.int32)
    counts = rng.randint(0, 6, total)
    counts[rng.choice(total, 4, replace=False)] = -1                    # cells without a list
    order = rng.permutation(total)                                       # pool filled in no particular order
    pool = [0]
    lists = np.zeros((total, 2), np.int64)
    wanted = {}
    for c in order:
        if counts[c] < 0:
            continue
        entries = (0x80000000 | rng.randint(0, 2**24, counts[c])).astype(np.int64)
        lists[c] = [len(pool), 0x80000000 | counts[c]]
        pool.extend(entries.tolist())
        wanted[c] = sorted(entries.tolist())
    pool[0] = len(pool) - 1
    to_i32 = lambda a: torch.tensor((np.asarray(a, np.int64) & 0xffffffff).astype(np.uint32).view(np.int32))
    # (next to every candidate its wall's row: here a row that names its entry, so that the repacking can be followed)
    rows = torch.tensor(np.asarray(pool, np.float64)[:, None]*np.array([1., 2., 3., 4.]) % 1000, dtype=torch.float32)
    lg = (vals, starts, geom, .25, int(cells.max()), to_i32(lists), to_i32(pool), rows)
    for start, stop in [(0, 2), (1, 4), (2, 3), (0, 4)]:
        v, s, g, cell, mx, l, p, pr = sharding._shard_light_grid(lg, start, stop, 'cpu')
        assert pr.shape == (len(p), 4)
        want_rows = torch.tensor(((p.long() & 0xffffffff).double().numpy()[:, None]*np.array([1., 2., 3., 4.])) % 1000, dtype=torch.float32)
        assert torch.equal(pr[1:], want_rows[1:])                          # every entry still has its own row beside it
        assert sharding._shard_light_grid(lg[:7], start, stop, 'cpu')[7] is None     # (a grid baked without rows shards without them)
        c0, c1 = int(starts[start]), int(starts[stop]) if stop < 4 else total
        assert torch.equal(v[:-1], vals[c0:c1]) and not v[-1].any() and torch.equal(g, geom[start:stop]) and cell == .25    # (+ the padding row)
        assert len(l) == len(v)
        assert s.tolist() == (starts[start:stop] - c0).tolist() and mx == int(cells[start:stop].max())
        l64, p64 = l.long() & 0xffffffff, p.long() & 0xffffffff
        assert int(p64[0]) == len(p) - 1
        for i, c in enumerate(range(c0, c1)):
            if counts[c] < 0:
                assert int(l64[i, 1]) == 0
            else:
                first, n = int(l64[i, 0]), int(l64[i, 1]) & 0x7fffffff
                assert int(l64[i, 1]) >> 31 == 1 and n == counts[c] and first >= 1
                assert sorted(p64[first:first + n].tolist()) == wanted[c]


@pytest.mark.parametrize('shared_stream', [False, True])
def test_vectorised_scenery_equals_the_reference_assembly_loop(shared_stream):
    """scene.scenery works per distinct floorplan and gathers; the result must be what the reference's per-env loop
    (scene.py:75-100) produces, value for value, random streams included - also when geometries repeat and when the
    wall patterns come from the same global stream as the light intensities."""
    from megastep_amd import scene, cubicasa, toys
    from tests import util
    pool = cubicasa.sample(3, n_unique=16)
    box = toys.box()
    geoms = [pool[0], box, pool[1], pool[0], pool[2], box, box, pool[1]]
    np.random.seed(11)
    want = util.scenery_by_the_book(geoms, 3, np.random if shared_stream else np.random.RandomState(4))
    np.random.seed(11)
    got = scene.scenery(geoms, 3, device='cpu', bake=False, random=np.random if shared_stream else np.random.RandomState(4))
    for k, t in [('lights_vals', got.lights.vals), ('lights_widths', got.lights.widths), ('lines_vals', got.lines.vals),
                 ('lines_widths', got.lines.widths), ('textures_vals', got.textures.vals), ('textures_widths', got.textures.widths)]:
        np.testing.assert_array_equal(t.numpy(), want[k], err_msg=k)
    # envs built from one geometry object name the first of them as their representative
    assert got.geom.tolist() == [0, 1, 2, 0, 4, 1, 1, 2]
    assert scene.scenery(pool, 1, device='cpu', bake=False).geom is None


def test_fast_scenery_has_the_same_layout_and_statistics():
    """fast=True draws intensities and wall patterns on the device: everything deterministic (lines, texel counts,
    which colour a texel has) is unchanged; brightness levels follow the same law (in [.5, 1), agent texels 1, level
    changes about every 10th texel); intensities are U(.5, 2)."""
    from megastep_amd import scene, cubicasa
    pool = cubicasa.sample(4, n_unique=16)
    geoms = [pool[i % 4] for i in range(12)]
    np.random.seed(0)
    ref = scene.scenery(geoms, 2, device='cpu', bake=False)
    torch.manual_seed(0)
    got = scene.scenery(geoms, 2, device='cpu', bake=False, fast=True)
    np.testing.assert_array_equal(got.lines.vals.numpy(), ref.lines.vals.numpy())
    np.testing.assert_array_equal(got.textures.widths.numpy(), ref.textures.widths.numpy())
    np.testing.assert_array_equal(got.lights.vals[:, :2].numpy(), ref.lights.vals[:, :2].numpy())
    inten = got.lights.vals[:, 2].numpy()
    assert inten.min() >= .5 and inten.max() <= 2. and abs(inten.mean() - 1.25) < .1
    agent_texels = 2*2*8
    for e in range(len(geoms)):
        t0 = int(ref.textures.starts[int(ref.lines.starts[e])])
        t1 = int(ref.textures.ends[int(ref.lines.ends[e]) - 1])
        tex_got, tex_ref = got.textures.vals[t0:t1].numpy(), ref.textures.vals[t0:t1].numpy()
        np.testing.assert_array_equal(tex_got[:agent_texels], tex_ref[:agent_texels])       # agents are drawn flat
        # same hue, texel by texel: got = colour*b1, ref = colour*b2 with b in [.5, 1)
        b = tex_got[agent_texels:].max(1)/np.maximum(tex_ref[agent_texels:].max(1), 1e-9)
        assert (b > .5/1.).all() and (b < 1./.5).all()
    # the pattern's statistics, read off one wall colour channel: jumps in ~10% of the steps within a long wall
    w = ref.textures.widths.numpy()
    longest = int(np.argmax(w))
    t0 = int(ref.textures.starts[longest])
    seg = got.textures.vals[t0:t0 + w[longest]].numpy().max(1)
    assert len(seg) > 50 and .02 < (np.diff(seg) != 0).mean() < .3


def test_shards_of_sceneries_with_shared_floorplans():
    """Envs that share a floorplan share a light grid; a shard re-elects representatives among its own envs and takes
    exactly their cells along."""
    from megastep_amd import sharding, scene
    pool = cubicasa.sample(3, n_unique=16)
    geoms = [pool[0], pool[1], pool[0], pool[2], pool[1], pool[1], pool[0]]
    full = scene.scenery(geoms, 2, device='cpu', random=np.random.RandomState(0), bake=False)
    assert full.geom.tolist() == [0, 1, 0, 3, 1, 1, 0]
    shard = sharding.shard_scenery(full, 1, 2)                # envs 4, 5, 6: floorplans 1, 1, 0
    assert shard.geom.tolist() == [0, 0, 2]
    assert sharding.shard_scenery(full, 0, 7).geom.tolist() == [0]
    # a light grid with shared cells: envs 0, 2, 6 -> cells [0, 6), envs 1, 4, 5 -> [6, 10), env 3 -> [10, 14)
    dims = torch.tensor([[3., 2.], [1., 4.], [3., 2.], [2., 2.], [1., 4.], [1., 4.], [3., 2.]])
    grid = torch.cat([torch.zeros(7, 2), dims], 1)
    starts = torch.tensor([0, 6, 0, 10, 6, 6, 0], dtype=torch.int32)
    vals = torch.arange(14*4, dtype=torch.int32).view(14, 4)
    lists = torch.zeros((14, 2), dtype=torch.int32)
    pool_ = torch.zeros(1, dtype=torch.int32)
    v, s, g, cell, mx, l, p, _ = sharding._shard_light_grid((vals, starts, grid, .25, 6, lists, pool_), 4, 7, 'cpu', shard.geom)
    assert s.tolist() == [0, 0, 4] and torch.equal(v[:-1], torch.cat([vals[6:10], vals[0:6]])) and torch.equal(g, grid[4:7])     # (v ends in a padding row)


def test_cost_balanced_env_slices():
    """Slices balanced by work (lines x agents x rays) instead of env count: contiguous, complete, non-empty, and no
    rank carries much more than its share."""
    rng = np.random.RandomState(0)
    for world in (2, 3, 8):
        cost = rng.randint(150, 1200, 4096).astype(float)
        slices = [sharding.env_slice(4096, r, world, cost) for r in range(world)]
        assert slices[0][0] == 0 and slices[-1][1] == 4096
        assert all(a[1] == b[0] for a, b in zip(slices, slices[1:])) and all(b > a for a, b in slices)
        loads = np.array([cost[a:b].sum() for a, b in slices])
        assert loads.max() <= cost.sum()/world + cost.max()
        by_count = np.array([cost[a:b].sum() for a, b in (sharding.env_slice(4096, r, world) for r in range(world))])
        assert loads.max() <= by_count.max() + cost.max()
    # degenerate weights still give every rank an env
    assert [sharding.env_slice(4, r, 4, [0, 0, 0, 9]) for r in range(4)] == [(0, 1), (1, 2), (2, 3), (3, 4)]
    assert [sharding.env_slice(5, r, 2, [9, 0, 0, 0, 0]) for r in range(2)] == [(0, 1), (1, 5)]
    with pytest.raises(ValueError):
        sharding.env_slice(3, 0, 4, [1, 1, 1])
    sc = scene.scenery(cubicasa.sample(6, n_unique=16), 2, device='cpu', bake=False)
    cost = sharding.render_cost(sc, 64)
    assert cost.tolist() == (sc.lines.widths.double()*2*64).tolist()
    parts = [sharding.shard_scenery(sc, r, 2, cost=cost) for r in range(2)]
    assert sum(len(p.lines) for p in parts) == 6
