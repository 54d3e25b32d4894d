"""Host-side logic against golden vectors recorded from the reference's own Python modules
(tests/golden/make_golden.py, run in the authoring container with the CUDA extension stubbed out)."""
import os
import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'reference_host.npz'))


def close(a, b, tol=1e-12):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=0, atol=tol)


def test_constants():
    from megastep_amd import core
    assert core.AGENT_WIDTH == G['AGENT_WIDTH'] and core.TEXTURE_RES == G['TEXTURE_RES']
    assert core.AGENT_RADIUS == G['AGENT_RADIUS']
    close(core.gamma_decode(G['gamma_decode_in']), G['gamma_decode_out'])
    close(core.gamma_encode(G['gamma_decode_in']), G['gamma_encode_out'])


def test_agent_model_and_colors():
    from megastep_amd import scene
    close(scene.agent_model(), G['agent_model'])
    close(scene.agent_colors(), G['agent_colors'])
    assert scene.agent_model().shape == (8, 2, 2)
    close(scene.agent_model()[0], [[-.0375, -.075], [.0375, -.075]])


def test_lengths_resolutions_pattern():
    from megastep_amd import scene
    close(scene.lengths(G['walls']), G['lengths'])
    np.testing.assert_array_equal(scene.resolutions(np.concatenate([scene.agent_model(), G['walls']])), G['resolutions'])
    close(scene.wall_pattern(500, random=np.random.RandomState(5)), G['wall_pattern'])


@pytest.mark.parametrize('n_agents', [1, 3])
def test_init_textures_reproduces_reference_rng_stream(n_agents):
    from megastep_amd import scene
    agentlines = np.tile(scene.agent_model(), (n_agents, 1, 1))
    agentcolors = np.tile(scene.agent_colors(), (n_agents, 1))
    for name, walls, seed in [('box', G['box_walls'], 1), ('walls', G['walls'], 2)]:
        tex, widths = scene.init_textures(agentlines, agentcolors, walls, np.random.RandomState(seed))
        np.testing.assert_array_equal(widths, G[f'{name}_texwidths_{n_agents}'])
        close(tex, G[f'{name}_textures_{n_agents}'])
    assert list(G['box_texwidths_1']) == [2]*8 + [100]*4


def test_random_lights():
    from megastep_amd import scene
    close(scene.random_lights(G['lights'], random=np.random.RandomState(11)), G['random_lights'])


def test_box_walls_match_reference_toys():
    from megastep_amd import toys
    close(toys.box().walls, G['box_walls'])
    close(toys.box().lights, [[3.5, 3.5]])


@pytest.mark.parametrize('name', ['small', 'big'])
def test_ragged_numpy(name):
    from megastep_amd import ragged
    r = ragged.RaggedNumpy(G[f'ragged_{name}_vals'], G[f'ragged_{name}_widths'])
    for k in ('starts', 'ends', 'inverse'):
        np.testing.assert_array_equal(getattr(r, k), G[f'ragged_{name}_{k}'])
    close(r[1], G[f'ragged_{name}_item1'])
    close(r[-1], G[f'ragged_{name}_itemlast'])
    close(r[1:3].vals, G[f'ragged_{name}_slice_vals'])
    np.testing.assert_array_equal(r[1:3].widths, G[f'ragged_{name}_slice_widths'])


@pytest.mark.parametrize('name', ['small', 'big'])
def test_ragged_torch_matches_the_numpy_one(name):
    from megastep_amd import ragged
    vals = torch.as_tensor(G[f'ragged_{name}_vals']).float()
    widths = torch.as_tensor(G[f'ragged_{name}_widths']).int()
    r = ragged.Ragged(vals, widths)
    for k in ('starts', 'ends', 'inverse'):
        assert getattr(r, k).dtype == torch.int32
        np.testing.assert_array_equal(getattr(r, k).numpy(), G[f'ragged_{name}_{k}'])
    close(r[1].numpy(), G[f'ragged_{name}_item1'], 1e-6)
    close(r[-1].numpy(), G[f'ragged_{name}_itemlast'], 1e-6)
    close(r[1:3].vals.numpy(), G[f'ragged_{name}_slice_vals'], 1e-6)
    np.testing.assert_array_equal(r[1:3].widths.numpy(), G[f'ragged_{name}_slice_widths'])


def test_geometry_helpers():
    from megastep_amd import geometry
    close(geometry.centers(G['centers_in'], (36, 41), .2), G['centers_out'])
    np.testing.assert_array_equal(geometry.indices(G['indices_in'], (36, 41), .2), G['indices_out'])
    close(geometry.centers(np.array([[0, 0]]), (36, 36), .2), G['centers_origin'])
    close(geometry.unique(G['unique_in']), G['unique_out'])
    corners = G['box_walls'][:, 0]
    close([geometry.signed_area(corners), geometry.signed_area(corners[::-1])], G['signed_area'], 1e-9)
    assert geometry.cyclic_pairs([1, 2, 3]) == [(1, 2), (2, 3), (3, 1)]


def test_frames():
    from megastep_amd import modules
    ang, vec = torch.as_tensor(G['frame_angles']), torch.as_tensor(G['frame_vectors'])
    close(modules.to_global_frame(ang, vec).numpy(), G['to_global'], 1e-7)
    close(modules.to_local_frame(ang, vec).numpy(), G['to_local'], 1e-7)


class _FakeCore:
    n_envs, n_agents, res, fps, device = 3, 2, 16, 10, 'cpu'

    def __init__(self):
        from megastep_amd import core
        self.agent_radius = core.AGENT_RADIUS
        self.random = np.random.RandomState(1)


def test_observation_modules():
    from megastep_amd import modules, arrdict
    fc = _FakeCore()
    dist, screen = torch.as_tensor(G['obs_distances']), torch.as_tensor(G['obs_screen'])
    r = arrdict.arrdict(distances=dist.unsqueeze(2), screen=screen.unsqueeze(2).permute(0, 1, 4, 2, 3))
    for sub in (1, 4):
        d = modules.Depth(fc, subsample=sub, max_depth=10)
        out = d(r)
        assert out.shape == (3, 2, 1, 1, 16//sub) and d.space.shape == (2, 1, 1, 16//sub)
        close(out.numpy(), G[f'depth_sub{sub}'], 1e-7)
        c = modules.RGB(fc, subsample=sub)
        out = c(r)
        assert out.shape == (3, 2, 3, 1, 16//sub)
        close(out.numpy(), G[f'rgb_sub{sub}'], 1e-7)
    close(modules.downsample(dist, 4).numpy(), G['downsample'])


def test_movement_modules(monkeypatch):
    from megastep_amd import modules, arrdict, cuda
    calls = []
    monkeypatch.setattr(cuda, 'physics', lambda scenery, agents: calls.append('physics'))
    fc = _FakeCore()
    fc.scenery = None
    fc.agents = arrdict.arrdict(
        angles=torch.as_tensor(G['move_angles']).clone(), positions=torch.zeros(3, 2, 2),
        angvelocity=torch.as_tensor(G['move_angvel0']).clone(), velocity=torch.as_tensor(G['move_vel0']).clone())
    decision = arrdict.arrdict(actions=torch.as_tensor(G['move_actions']))
    mom = modules.MomentumMovement(fc, accel=5, ang_accel=180, decay=.125)
    close(mom._actionset.velocity.numpy(), G['momentum_actionset_v'], 1e-7)
    close(mom._actionset.angvelocity.numpy(), G['momentum_actionset_w'], 1e-7)
    mom(decision)
    close(fc.agents.velocity.numpy(), G['momentum_vel'], 1e-6)
    close(fc.agents.angvelocity.numpy(), G['momentum_angvel'], 1e-6)
    modules.SimpleMovement(fc, speed=10, ang_speed=180)(decision)
    close(fc.agents.velocity.numpy(), G['simple_vel'], 1e-6)
    close(fc.agents.angvelocity.numpy(), G['simple_angvel'], 1e-6)
    assert calls == ['physics', 'physics']
    close(modules.IMU(fc)().numpy(), G['imu'], 1e-7)
    assert mom.space.shape == (2, 7)


def test_spawn_tables_reproduce_reference_rng_stream():
    from megastep_amd import modules, arrdict
    mask = G['spawn_mask']
    geoms = [arrdict.arrdict(masks=mask, res=.2), arrdict.arrdict(masks=mask.T.copy(), res=.2)]
    np.random.seed(4)
    close(modules.random_empty_positions(geoms, 2, 10), G['spawn_positions'])
    fc = _FakeCore()
    fc.n_envs = 2
    np.random.seed(4)
    sp = modules.RandomSpawns(geoms, fc, n_spawns=10)
    close(sp._spawns.angles.numpy(), G['spawns_angles'], 1e-5)
    close(sp._spawns.positions.numpy(), G['spawns_positions'], 1e-6)


def test_masks_reproduce_the_reference_docs_figure():
    """geometry.masks on the geometry tutorial's 5 m box against what the reference's own masks() returned for it, read
    off docs/tutorials/geometry/walls-masks.png cell by cell (tests/golden/make_docs_images.py): the reference uses
    rasterio + shapely for this, absent here, so this figure is the only pin of our rasteriser's conventions."""
    import os
    from megastep_amd import geometry
    want = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'docs_geometry_masks.npy'))
    corners = 5*np.array([[0, 0], [0, 1], [1, 1], [1, 0]]) + 1
    walls = np.stack(geometry.cyclic_pairs(corners))
    got = geometry.masks(walls, [corners])
    assert got.shape == want.shape == (36, 36)
    np.testing.assert_array_equal(got, want)
    # and toys.box() is that geometry (toys.py:5-16)
    from megastep_amd import toys
    np.testing.assert_array_equal(toys.box().masks, want)
