"""Shared helpers for the parity tests: moving a product-side Core into the oracle's numpy world and comparing."""
import numpy as np
import torch
from oracle import oracle as O


def scene_dict(scenery):
    """cuda.Scenery -> the oracle's scene dict (numpy copies)."""
    n = lambda t: t.detach().cpu().numpy().copy()
    return dict(
        n_agents=scenery.n_agents, model=n(scenery.model),
        lights_vals=n(scenery.lights.vals), lights_widths=n(scenery.lights.widths),
        lines_vals=n(scenery.lines.vals), lines_widths=n(scenery.lines.widths),
        textures_vals=n(scenery.textures.vals), textures_widths=n(scenery.textures.widths),
        baked_vals=n(scenery.baked.vals))


def agents_dict(agents):
    n = lambda t: t.detach().cpu().numpy().copy()
    return dict(angles=n(agents.angles), positions=n(agents.positions),
                angvelocity=n(agents.angvelocity), velocity=n(agents.velocity))


class OracleWorld:
    """The oracle's copy of a Core: same scene, same config, agents pulled on demand."""

    def __init__(self, core):
        self.scene = O.Scene(scene_dict(core.scenery))
        self.cfg = O.config(core.agent_radius, core.res, core.fov, core.fps)
        self.agents = agents_dict(core.agents)

    def pull_agents(self, core):
        self.agents = agents_dict(core.agents)

    def pull_baked(self, core):
        self.scene.baked_vals[:] = core.scenery.baked.vals.cpu().numpy()

    def bake(self):
        return O.bake(self.scene, self.cfg).copy()

    def physics(self):
        progress, self.agents = O.physics(self.scene, self.agents, self.cfg)
        return progress, self.agents

    def render(self):
        return O.render(self.scene, self.agents, self.cfg)


def spawn(core, geometries, seed=0):
    """Puts every agent on a random free cell of its geometry with a random heading."""
    rng = np.random.RandomState(seed)
    from megastep_amd import geometry
    pos = np.zeros((core.n_envs, core.n_agents, 2), np.float32)
    for e, g in enumerate(geometries):
        free = np.stack((g['masks'] > 0).nonzero(), -1)
        pick = free[rng.choice(len(free), core.n_agents)]
        pos[e] = geometry.centers(pick, g['masks'].shape, g['res'])
    core.agents.positions[:] = torch.as_tensor(pos, device=core.device)
    core.agents.angles[:] = torch.as_tensor(rng.uniform(-180, 180, (core.n_envs, core.n_agents)).astype(np.float32), device=core.device)


def random_velocities(core, rng, speed=3., spin=180.):
    shape = (core.n_envs, core.n_agents)
    core.agents.velocity[:] = torch.as_tensor(rng.uniform(-speed, speed, shape + (2,)).astype(np.float32), device=core.device)
    core.agents.angvelocity[:] = torch.as_tensor(rng.uniform(-spin, spin, shape).astype(np.float32), device=core.device)


def _np(t):
    return t.detach().cpu().numpy()


def assert_physics_matches(core, p, progress_ref, agents_ref, atol=1e-5):
    """Bit-exact collision masks; float state within the north-star tolerance (1e-5)."""
    progress = _np(p.progress)
    np.testing.assert_array_equal(progress < 1, progress_ref < 1, err_msg='collision masks differ')
    np.testing.assert_allclose(progress, progress_ref, rtol=0, atol=atol)
    for k in ('positions', 'velocity', 'angvelocity'):
        np.testing.assert_allclose(_np(getattr(core.agents, k)), agents_ref[k], rtol=0, atol=atol, err_msg=k)
    # angles live in [-180, 180): one binary32 ulp there is 1.5e-5, so compare modulo that
    np.testing.assert_allclose(_np(core.agents.angles), agents_ref['angles'], rtol=0, atol=2e-5, err_msg='angles')


def assert_render_matches(core, r, ref, atol=1e-5):
    """Bit-exact hit indices (and miss sentinels); floats within 1e-5."""
    idx = _np(r.indices)
    np.testing.assert_array_equal(idx, ref['indices'], err_msg='hit indices differ')
    hit = idx >= 0
    for k in ('locations', 'dots', 'distances'):
        got, want = _np(getattr(r, k)), ref[k]
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want), err_msg=k)
        np.testing.assert_array_equal(np.isinf(got), np.isinf(want), err_msg=k)
        np.testing.assert_allclose(got[hit], want[hit], rtol=0, atol=atol, err_msg=k)
    np.testing.assert_allclose(_np(r.screen), ref['screen'], rtol=0, atol=atol, err_msg='screen')


def exact_fraction(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(((a == b) | (np.isnan(a) & np.isnan(b))).mean())


def scenery_by_the_book(geometries, n_agents, random=np.random):
    """The reference's assembly loop (scene.py:75-100) over the product's per-env helpers - repeat, gamma-decode and
    concatenate, one geometry at a time. Slow and obviously right: what scene.scenery's vectorised build is checked
    against. Returns numpy arrays (float32 / int32, as arrdict.torchify would make them)."""
    from megastep_amd import scene
    model = scene.agent_model()
    agentlines, agentcolors = np.tile(model, (n_agents, 1, 1)), np.tile(scene.agent_colors(), (n_agents, 1))
    lights, lines, texels, counts = [], [], [], []
    for g in geometries:
        lights.append(scene.random_lights(g['lights']))
        tex, cnt = scene.init_textures(agentlines, agentcolors, g['walls'], random)
        lines.append(np.concatenate([agentlines, g['walls']]))
        texels.append(tex)
        counts.append(cnt)
    return dict(
        lights_vals=np.concatenate(lights).astype(np.float32), lights_widths=np.array([len(x) for x in lights], np.int32),
        lines_vals=np.concatenate(lines).astype(np.float32), lines_widths=np.array([len(x) for x in lines], np.int32),
        textures_vals=np.concatenate(texels).astype(np.float32), textures_widths=np.concatenate(counts).astype(np.int32))


class OracleSubset:
    """The oracle's copy of a few envs of a (large) Core: what the full-size tests compare a sample against."""

    def __init__(self, core, envs):
        sc = core.scenery
        self.envs = np.asarray(envs)
        e = torch.as_tensor(self.envs, device=core.device)
        ln, li, tx = sc.lines, sc.lights, sc.textures
        rows = lambda r, which: torch.cat([torch.arange(int(r.starts[i]), int(r.ends[i]), device=core.device) for i in which]) \
            if len(which) else torch.zeros(0, dtype=torch.long, device=core.device)
        self.line_rows = rows(ln, self.envs)
        light_rows = rows(li, self.envs)
        tstart, tend = tx.starts.long()[self.line_rows], tx.ends.long()[self.line_rows]
        self.texel_rows = torch.cat([torch.arange(int(a), int(b), device=core.device) for a, b in
                                     zip(tx.starts.long()[ln.starts.long()[e]].tolist(), tx.ends.long()[ln.ends.long()[e] - 1].tolist())])
        assert int((tend - tstart).sum()) == len(self.texel_rows)
        n = lambda t: t.detach().cpu().numpy().copy()
        self.scene = O.Scene(dict(
            n_agents=sc.n_agents, model=n(sc.model),
            lights_vals=n(li.vals[light_rows]), lights_widths=n(li.widths[e]),
            lines_vals=n(ln.vals[self.line_rows]), lines_widths=n(ln.widths[e]),
            textures_vals=n(tx.vals[self.texel_rows]), textures_widths=n(tx.widths[self.line_rows]),
            baked_vals=n(sc.baked.vals[self.texel_rows])))
        self.cfg = O.config(core.agent_radius, core.res, core.fov, core.fps)
        self.pull_agents(core)

    def pull_agents(self, core):
        self.agents = {k: v[self.envs] for k, v in agents_dict(core.agents).items()}

    def bake(self):
        return O.bake(self.scene, self.cfg).copy()

    def physics(self):
        progress, self.agents = O.physics(self.scene, self.agents, self.cfg)
        return progress, self.agents

    def render(self):
        return O.render(self.scene, self.agents, self.cfg)


class _Rows:
    """A Core-like view of a few envs of a step's results, for the assert_* helpers above."""

    def __init__(self, core, envs):
        class A:
            pass
        self.agents = A()
        for k in ('angles', 'positions', 'angvelocity', 'velocity'):
            setattr(self.agents, k, getattr(core.agents, k)[envs])


def assert_subset_matches(core, sub, p, r):
    """Physics and render results of the full batch, at the sampled envs, against the oracle."""
    e = torch.as_tensor(sub.envs, device=core.device)
    prog_ref, agents_ref = sub.physics()

    class P:
        progress = p.progress[e]
    assert_physics_matches(_Rows(core, e), P, prog_ref, agents_ref)

    class R:
        pass
    for k in ('indices', 'locations', 'dots', 'distances', 'screen'):
        setattr(R, k, getattr(r, k)[e])
    assert_render_matches(None, R, sub.render())


def obstructed(I, C, walls):
    """obstructed() of kernels.cu:253-257 in float32 numpy: (n_points, n_walls) booleans."""
    f = np.float32
    a, v = walls[None, :, 0], (walls[:, 1] - walls[:, 0])[None]
    U = (C - I)[:, None]
    d = U[..., 0]*v[..., 1] - U[..., 1]*v[..., 0]
    PQ = a - I
    with np.errstate(divide='ignore', invalid='ignore'):
        s = (PQ[..., 0]*v[..., 1] - PQ[..., 1]*v[..., 0])/d
        t = (PQ[..., 0]*U[..., 1] - PQ[..., 1]*U[..., 0])/d
    return (np.abs(d) >= f(1e-3)) & (t > 0) & (t < 1) & (s > 0) & (s < f(.999))
