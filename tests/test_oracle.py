"""Pins the CPU oracle (oracle/megastep_oracle.c) before anything is compared against it.

What the reference offers for the hot path: one documented known answer for physics
(docs/tutorials/minimal-env/index.rst:140-145), the Ragged test vector (ragged.py:77-103, docs/concepts.rst:205-219),
and nothing for render/bake - those are pinned here by closed forms derived from the kernel semantics."""
import numpy as np
import pytest
import torch


R_AGENT = .15/2**.5


def box_scene(n_agents=1, n_envs=1, lights=None):
    from megastep_amd import scene, toys
    g = toys.box()
    lines = np.concatenate([np.tile(scene.agent_model(), (n_agents, 1, 1)), g.walls])
    texw = scene.resolutions(lines)
    lights = np.array([[3.5, 3.5, 1.]]) if lights is None else lights
    return dict(
        n_agents=n_agents, model=scene.agent_model(),
        lights_vals=np.tile(lights, (n_envs, 1)), lights_widths=[len(lights)]*n_envs,
        lines_vals=np.tile(lines, (n_envs, 1, 1)), lines_widths=[len(lines)]*n_envs,
        textures_vals=np.full((n_envs*texw.sum(), 3), .5), textures_widths=np.tile(texw, n_envs))


def agents(pos, ang=0., vel=(0., 0.), angvel=0.):
    pos = np.asarray(pos, np.float32).reshape(1, -1, 2)
    A = pos.shape[1]
    return dict(angles=np.full((1, A), ang, np.float32), positions=pos,
                angvelocity=np.full((1, A), angvel, np.float32), velocity=np.tile(np.asarray(vel, np.float32), (1, A, 1)))


def test_docs_known_answer(oracle):
    """box(5), agent at (3, 3), velocity (1000, 0), fps 10 -> (5.8649, 3.0000)."""
    S = oracle.Scene(box_scene())
    prog, new = oracle.physics(S, agents([[3., 3.]], vel=(1000., 0.)), oracle.config(R_AGENT, 64, 130, 10))
    np.testing.assert_allclose(new['positions'][0, 0], [5.8649, 3.0], atol=5e-5)
    # closed form from kernels.cu:143-146: 0.99 (1 - 1.001 R / 3) * 0.03 of a 100 m step
    np.testing.assert_allclose(prog[0, 0], .99*(1 - 1.001*R_AGENT/3)*.03, rtol=1e-6)
    assert (new['velocity'] == 0).all()


def test_free_motion_and_angle_wrap(oracle):
    S = oracle.Scene(box_scene())
    prog, new = oracle.physics(S, agents([[3., 3.]], ang=170., vel=(1., 2.), angvel=150.), oracle.config(R_AGENT, 64, 130, 10))
    assert prog[0, 0] == 1
    np.testing.assert_allclose(new['positions'][0, 0], [3.1, 3.2], atol=1e-6)
    np.testing.assert_allclose(new['angles'][0, 0], -175., atol=1e-4)       # 170 + 15 wraps to -175
    np.testing.assert_allclose(new['velocity'][0, 0], [1., 2.])            # untouched when nothing is hit


def test_normalize_degrees_is_aten_remainder(oracle):
    """kernels.cu:173-175 uses ATen `%` == torch.remainder; pin the oracle's to torch's own CPU kernel."""
    a = torch.cat([torch.linspace(-1000, 1000, 4001), torch.tensor([180., -180., 179.99998, 360., 540., -540., 0.])])
    want = (((a % 360.) + 180.) % 360.) - 180.
    got = np.array([oracle.lib().oracle_normalize_degrees(float(x)) for x in a], np.float32)
    np.testing.assert_array_equal(got, want.numpy())
    assert ((got >= -180) & (got < 180)).all()


def test_agent_agent_collision(oracle):
    """Two agents heading at each other stop short of touching (kernels.cu:119-133): r = 1.001 * 2 R."""
    S = oracle.Scene(box_scene(n_agents=2))
    ag = agents([[3., 3.5], [4., 3.5]])
    ag['velocity'] = np.array([[[10., 0.], [-10., 0.]]], np.float32)        # 1 m each per step: they would swap
    prog, new = oracle.physics(S, ag, oracle.config(R_AGENT, 64, 130, 10))
    r = 1.001*2*R_AGENT
    want = .99*(1. - r)/2.                                                   # relative speed 2 m/step closes 1 - r
    np.testing.assert_allclose(prog[0], [want, want], rtol=1e-5)
    gap = new['positions'][0, 1, 0] - new['positions'][0, 0, 0]
    assert gap > r and gap < r + .03


def test_analytic_box_raycast(oracle):
    """Centre of box(5), heading +x, fov 130, 8 rays: every ray meets a wall at 2.5 |ru| / max(1, |y|)."""
    S = oracle.Scene(box_scene())
    cfg = oracle.config(R_AGENT, 8, 130, 10)
    r = oracle.render(S, agents([[3.5, 3.5]]), cfg)
    y = (8 - 2*np.arange(8) - 1)*np.tan(np.deg2rad(65.))/8
    want = 2.5*np.sqrt(1 + y**2)/np.maximum(1, np.abs(y))
    np.testing.assert_allclose(r['distances'][0, 0], want, rtol=2e-6)
    np.testing.assert_allclose(r['distances'][0, 0], [2.83285, 3.11915, 3.20812, 2.58826, 2.58826, 3.20812, 3.11915, 2.83285], atol=1e-5)
    # walls follow the 8 agent lines: 8 top (y=6), 9 left, 10 bottom (y=1), 11 right (x=6)
    np.testing.assert_array_equal(r['indices'][0, 0], [8, 8, 11, 11, 11, 11, 10, 10])
    # rays are mirror images about the heading: locations mirror, dots flip sign
    np.testing.assert_allclose(r['locations'][0, 0, 2:6] + r['locations'][0, 0, 2:6][::-1], 1., atol=1e-6)
    np.testing.assert_allclose(r['dots'][0, 0], -r['dots'][0, 0][::-1], atol=1e-6)
    # render rewrote the agent's model lines around (3.5, 3.5) (kernels.cu:316-317)
    from megastep_amd import scene
    np.testing.assert_allclose(S.lines_vals[:8], scene.agent_model() + 3.5, atol=1e-6)


def test_rotated_view_and_miss_sentinels(oracle):
    """Heading 90 deg sees the top wall; an agent outside the box looking away sees nothing."""
    S = oracle.Scene(box_scene())
    cfg = oracle.config(R_AGENT, 8, 90, 10)
    r = oracle.render(S, agents([[3.5, 3.5]], ang=90.), cfg)
    assert (r['indices'][0, 0] == 8).all()
    y = (8 - 2*np.arange(8) - 1)*np.tan(np.deg2rad(45.))/8
    np.testing.assert_allclose(r['distances'][0, 0], 2.5*np.sqrt(1 + y**2), rtol=2e-6)
    r = oracle.render(S, agents([[8., 3.5]], ang=0.), cfg)
    assert (r['indices'] == -1).all() and np.isnan(r['locations']).all() and np.isnan(r['dots']).all()
    assert np.isposinf(r['distances']).all() and (r['screen'] == 0).all()


def test_bake_closed_form(oracle):
    """Unoccluded texel lighting: min(1, 0.1 + 2 I / max(d^2, 1)) (kernels.cu:238-268); agent texels sit at the
    origin outside the box, so every wall shadows them and they get the ambient 0.1."""
    S = oracle.Scene(box_scene(lights=np.array([[3.5, 3.5, 1.5]])))
    baked = oracle.bake(S, oracle.config(R_AGENT, 8, 130, 10)).copy()
    np.testing.assert_allclose(baked[:16], .1, atol=1e-7)
    # wall 8 runs (6,6) -> (1,6), 100 texels; texel k sits at x = 6 - 5 (k + .5)/100
    k = np.arange(100)
    x = 6 - 5*(k + .5)/100
    d2 = (x - 3.5)**2 + 2.5**2
    np.testing.assert_allclose(baked[16:116], np.minimum(1, .1 + 2*1.5/np.maximum(d2, 1)), rtol=2e-6)


def test_shadow_from_a_column(oracle):
    """A column between a light and the far wall leaves a shadow of ambient-only texels."""
    from megastep_amd import scene, toys
    g = toys.column()
    walls = np.concatenate([g.walls, toys.box().walls])
    lines = np.concatenate([scene.agent_model(), walls])
    texw = scene.resolutions(lines)
    sc = dict(n_agents=1, model=scene.agent_model(), lights_vals=np.array([[2.5, 3.5, 1.]]), lights_widths=[1],
              lines_vals=lines, lines_widths=[len(lines)], textures_vals=np.full((texw.sum(), 3), .5), textures_widths=texw)
    S = oracle.Scene(sc)
    baked = oracle.bake(S, oracle.config(R_AGENT, 8, 130, 10))
    right = baked[S.textures_starts[8 + 4 + 3]:][:100]        # wall (6,1)->(6,6), texel k at y = 1 + 5 (k+.5)/100
    y = 1 + 5*(np.arange(100) + .5)/100
    shadow = np.abs(y - 3.5) < .05*(3.5/1.0) - .03             # column half-width .05 at 1.0 m, wall at 3.5 m
    lit = np.abs(y - 3.5) > .05*(3.5/.95) + .06
    assert np.allclose(right[shadow], .1) and (right[lit] > .1).all()


def test_texture_filter(oracle):
    """kernels.cu:394-405, literally: two taps, weights from distances to l+1 and r+1 (+1e-3)."""
    import ctypes as C
    def f(x, w):
        l, r, lw, rw = C.c_int(), C.c_int(), C.c_float(), C.c_float()
        oracle.lib().oracle_filter(x, w, C.byref(l), C.byref(r), C.byref(lw), C.byref(rw))
        return l.value, r.value, lw.value, rw.value
    assert f(0., 10)[:2] == (0, 0)
    l, r, lw, rw = f(.5, 10)            # y = 5.5 -> l = 4, r = 5
    assert (l, r) == (4, 5) and abs(lw + rw - 1) < 1e-6
    np.testing.assert_allclose(lw, .501/(1.002), rtol=1e-5)
    assert f(1., 10)[:2] == (8, 9)      # y clamps to w - 1
    assert f(.3, 1)[:2] == (0, 0)


def test_sincospi_against_double(oracle):
    xs = np.concatenate([np.linspace(-4, 4, 20001), np.random.RandomState(0).uniform(-1000, 1000, 5000)]).astype(np.float32)
    got = np.array([oracle.sincospi(x) for x in xs], np.float64)
    red = np.float64(xs) % 2.0                                   # exact: float32 values, double remainder
    want = np.stack([np.sin(np.pi*red), np.cos(np.pi*red)], 1)
    ulp = np.spacing(np.maximum(np.abs(want), 2.**-24).astype(np.float32)).astype(np.float64)
    assert (np.abs(got - want) <= .5000001*ulp + 1e-12).all()
    for x, s, c in [(0., 0., 1.), (.5, 1., 0.), (1., 0., -1.), (1.5, -1., 0.), (-.5, -1., 0.), (2., 0., 1.)]:
        assert oracle.sincospi(x) == (np.float32(s), np.float32(c))


def test_ragged_index(oracle):
    """ragged.py:93-103 / docs/concepts.rst:205-219 + a zero-width row (defined here, undefined in the reference)."""
    starts, ends, inverse = oracle.ragged_index([3, 1, 2])
    assert list(starts) == [0, 3, 4] and list(ends) == [3, 4, 6] and list(inverse) == [0, 0, 0, 1, 2, 2]
    starts, ends, inverse = oracle.ragged_index([2, 0, 3, 0])
    assert list(starts) == [0, 2, 2, 5] and list(ends) == [2, 2, 5, 5] and list(inverse) == [0, 0, 2, 2, 2]


def test_hysteresis_keeps_first_of_coincident_walls(oracle):
    """The 1e-4 z-fight rule (kernels.cu:369): of two walls at the same depth the EARLIER index wins, and a later
    wall only wins by being closer by more than 1e-4."""
    from megastep_amd import scene
    def first_hit(walls):
        lines = np.concatenate([scene.agent_model(), walls])
        texw = scene.resolutions(lines)
        sc = dict(n_agents=1, model=scene.agent_model(), lights_vals=np.zeros((0, 3)), lights_widths=[0],
                  lines_vals=lines, lines_widths=[len(lines)], textures_vals=np.full((texw.sum(), 3), .5), textures_widths=texw)
        r = oracle.render(oracle.Scene(sc), agents([[2., 2.]]), oracle.config(R_AGENT, 1, 90, 10))
        return r['indices'][0, 0, 0], r['distances'][0, 0, 0]
    wall = lambda x: np.array([[[x, 1.], [x, 3.]]])
    assert first_hit(np.concatenate([wall(4.), wall(4.)]))[0] == 8
    assert first_hit(np.concatenate([wall(4.), wall(3.99995)]))[0] == 8       # closer by 5e-5: not enough
    assert first_hit(np.concatenate([wall(4.), wall(3.9995)]))[0] == 9        # closer by 5e-4: wins
    assert first_hit(np.concatenate([wall(3.99995), wall(4.)]))[0] == 8


def test_render_against_the_reference_docs_image(oracle):
    """The one rendered output the reference publishes: docs/tutorials/minimal-env/render.png = gamma_encode(r.screen)
    for toys.box(), agent at (3, 3), heading 0, 64 rays, fov 130 (index.rst:100-120; columns extracted by
    tests/golden/make_docs_images.py). Its texture pattern and light intensity were random, so brightness is not
    comparable - but which wall each of the 64 rays lands on is, and so is each wall's hue, which survives any scalar
    brightness factor through the gamma curve: (k c^2.2)^(1/2.2) = k' c."""
    import os
    from megastep_amd import core, scene, toys
    cols = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'docs_render_columns.npy')).astype(np.float64)
    assert cols.shape == (64, 3)
    # the image's own segmentation: purple (R > B > G) | brown (R > G > B) | blue (B > G > R)
    order = np.argsort(-cols, 1)
    classes = np.array([{(0, 2, 1): 0, (0, 1, 2): 1, (2, 1, 0): 2}[tuple(o)] for o in order])
    assert (classes[:17] == 0).all() and (classes[17:42] == 1).all() and (classes[42:] == 2).all()

    g = toys.box()
    agentlines = scene.agent_model()
    lines = np.concatenate([agentlines, g.walls])
    tex, texw = scene.init_textures(agentlines, scene.agent_colors(), g.walls, np.random.RandomState(0))
    S = oracle.Scene(dict(
        n_agents=1, model=agentlines, lights_vals=scene.random_lights(g.lights, np.random.RandomState(1)), lights_widths=[len(g.lights)],
        lines_vals=lines, lines_widths=[len(lines)], textures_vals=tex, textures_widths=texw))
    cfg = oracle.config(core.AGENT_RADIUS, 64, 130, 10)
    S.baked_vals[:] = oracle.bake(S, cfg)
    r = oracle.render(S, agents([[3., 3.]]), cfg)
    idx = r['indices'][0, 0]
    # wall 0 (y = 6, on the agent's left) for 17 rays, wall 3 (x = 6, ahead) for 25, wall 2 (y = 1, right) for 22
    np.testing.assert_array_equal(idx, [8]*17 + [11]*25 + [10]*22)
    shown = core.gamma_encode(r['screen'][0, 0].astype(np.float64))
    cos = (shown*cols).sum(1)/np.linalg.norm(shown, axis=1)/np.linalg.norm(cols, axis=1)
    assert cos.min() > .999, cos.min()                     # hue of every column = the reference's, to 8-bit rounding


@pytest.mark.parametrize('n_agents,res,fov', [(1, 32, 130), (3, 64, 70)])
def test_pure_pytorch_step_matches_the_c_oracle(oracle, n_agents, res, fov):
    """oracle/torch_step.py (bench.py's pure-PyTorch CPU baseline) against the C restatement: same collision masks and
    hit indices, floats within 1e-5, over a few steps with agents meeting walls and each other."""
    import torch
    from oracle import torch_step
    from megastep_amd import core, cubicasa, scene
    from tests import util
    np.random.seed(0)
    geoms = cubicasa.sample(5, n_unique=16)
    sc = scene.scenery(geoms, n_agents, device='cpu', random=np.random.RandomState(0), bake=False)
    c = core.Core(sc, res=res, fov=fov)
    util.spawn(c, geoms, seed=2)
    if n_agents > 1:                                       # two agents face to face so that rays land on an agent
        c.agents.positions[0, 1] = c.agents.positions[0, 0] + torch.tensor([.6, 0.])
        c.agents.angles[0, 0], c.agents.angles[0, 1] = 0., 180.
    ref = util.OracleWorld(c)
    ref.bake()
    world = torch_step.World({k: getattr(ref.scene, k) for k in
                              ('n_agents', 'model', 'lines_vals', 'lines_widths', 'lights_vals', 'lights_widths',
                               'textures_vals', 'textures_widths', 'baked_vals')}, c.agent_radius, res, fov, c.fps)
    rng = np.random.RandomState(1)
    dyn = 0
    for step in range(3):
        util.random_velocities(c, rng, speed=3. if step else 0.)
        ref.pull_agents(c)
        agents = {k: torch.as_tensor(v.copy()) for k, v in ref.agents.items()}
        progress, frame = torch_step.step(world, agents)
        want_p, want_agents = ref.physics()
        want = ref.render()
        np.testing.assert_array_equal(progress.numpy() < 1, want_p < 1)
        np.testing.assert_allclose(progress.numpy(), want_p, rtol=0, atol=1e-5)
        for k in ('positions', 'velocity', 'angvelocity'):
            np.testing.assert_allclose(agents[k].numpy(), want_agents[k], rtol=0, atol=1e-5, err_msg=k)
        np.testing.assert_allclose(agents['angles'].numpy(), want_agents['angles'], rtol=0, atol=2e-5)
        np.testing.assert_array_equal(frame['indices'].numpy(), want['indices'])
        hit = want['indices'] >= 0
        for k in ('locations', 'dots', 'distances'):
            np.testing.assert_allclose(frame[k].numpy()[hit], want[k][hit], rtol=0, atol=1e-5, err_msg=k)
        np.testing.assert_allclose(frame['screen'].numpy(), want['screen'], rtol=0, atol=1e-5)
        dyn += int(((want['indices'] >= 0) & (want['indices'] < 8*n_agents)).sum())
        for k in ('angles', 'positions', 'velocity', 'angvelocity'):           # carry on from the oracle's state
            getattr(c.agents, k)[:] = torch.as_tensor(want_agents[k])
    assert n_agents == 1 or dyn > 0
