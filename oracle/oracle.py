"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE -- see megastep_oracle.c header).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module. It works on plain numpy arrays and knows nothing about the product package.

A *scene* here is a dict of numpy arrays mirroring the reference's ``Scenery`` (src/common.h:179-214)::

    n_agents         int
    model            (M, 2, 2) f32
    lights_vals      (sum I, 3) f32      lights_widths   (N,) i32
    lines_vals       (sum L, 2, 2) f32   lines_widths    (N,) i32
    textures_vals    (sum T, 3) f32      textures_widths (sum L,) i32
    baked_vals       (sum T,) f32

and *agents* is a dict with ``angles (N,A)``, ``positions (N,A,2)``, ``angvelocity (N,A)``, ``velocity (N,A,2)``.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libmegastep_oracle.so')

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int)


class OrScenery(C.Structure):
    _fields_ = [
        ('n_envs', C.c_int), ('n_agents', C.c_int), ('n_model', C.c_int),
        ('lights_vals', _f32p), ('lights_widths', _i32p), ('lights_starts', _i32p),
        ('lines_vals', _f32p), ('lines_widths', _i32p), ('lines_starts', _i32p), ('lines_inverse', _i32p),
        ('textures_vals', _f32p), ('textures_widths', _i32p), ('textures_starts', _i32p), ('textures_inverse', _i32p),
        ('model', _f32p), ('baked_vals', _f32p),
        ('n_lines_total', C.c_int), ('n_lights_total', C.c_int), ('n_texels_total', C.c_int)]


class OrAgents(C.Structure):
    _fields_ = [('angles', _f32p), ('positions', _f32p), ('angvelocity', _f32p), ('velocity', _f32p)]


class OrRender(C.Structure):
    _fields_ = [('indices', _i32p), ('locations', _f32p), ('dots', _f32p), ('distances', _f32p), ('screen', _f32p)]


class OrConfig(C.Structure):
    _fields_ = [('agent_radius', C.c_float), ('res', C.c_int), ('fov', C.c_float), ('fps', C.c_float)]


def build(force=False):
    """Compiles the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, 'megastep_oracle.c')
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libmegastep_oracle.so'], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_physics.argtypes = [C.POINTER(OrScenery), C.POINTER(OrAgents), _f32p, C.POINTER(OrConfig)]
        _lib.oracle_render.argtypes = [C.POINTER(OrScenery), C.POINTER(OrAgents), C.POINTER(OrRender), C.POINTER(OrConfig)]
        _lib.oracle_bake.argtypes = [C.POINTER(OrScenery), C.POINTER(OrConfig)]
        _lib.oracle_ragged_index.argtypes = [_i32p, C.c_int, _i32p, _i32p, _i32p]
        _lib.oracle_sincospi.argtypes = [C.c_float, _f32p, _f32p]
        _lib.oracle_collision_cs.argtypes = [C.c_float]*9
        _lib.oracle_collision_cs.restype = C.c_float
        _lib.oracle_collision_cc.argtypes = [C.c_float]*9
        _lib.oracle_collision_cc.restype = C.c_float
        _lib.oracle_normalize_degrees.argtypes = [C.c_float]
        _lib.oracle_normalize_degrees.restype = C.c_float
        _lib.oracle_filter.argtypes = [C.c_float, C.c_int, _i32p, _i32p, _f32p, _f32p]
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(_f32p if a.dtype == np.float32 else _i32p)


def ragged_index(widths):
    """starts, ends, inverse of a ragged with these widths (src/common.h:91-128)."""
    widths = _i32(widths)
    starts, ends = np.zeros_like(widths), np.zeros_like(widths)
    inverse = np.zeros(int(widths.sum()), dtype=np.int32)
    rc = lib().oracle_ragged_index(_p(widths), len(widths), _p(starts), _p(ends), _p(inverse))
    assert rc == 0, 'negative width'
    return starts, ends, inverse


def sincospi(x):
    s, c = C.c_float(), C.c_float()
    lib().oracle_sincospi(np.float32(x), C.byref(s), C.byref(c))
    return np.float32(s.value), np.float32(c.value)


def config(agent_radius, res, fov, fps):
    return OrConfig(float(agent_radius), int(res), float(fov), float(fps))


class Scene:
    """Owns contiguous copies of a scene dict and the matching OrScenery struct."""

    def __init__(self, scene):
        self.n_agents = int(scene['n_agents'])
        self.model = _f32(scene['model'])
        self.lights_vals = _f32(scene['lights_vals']).reshape(-1, 3).copy()
        self.lights_widths = _i32(scene['lights_widths']).copy()
        self.lines_vals = _f32(scene['lines_vals']).reshape(-1, 2, 2).copy()
        self.lines_widths = _i32(scene['lines_widths']).copy()
        self.textures_vals = _f32(scene['textures_vals']).reshape(-1, 3).copy()
        self.textures_widths = _i32(scene['textures_widths']).copy()
        n_tex = self.textures_vals.shape[0]
        baked = scene.get('baked_vals')
        self.baked_vals = np.ones(n_tex, np.float32) if baked is None else _f32(baked).copy()
        self.lights_starts, _, _ = ragged_index(self.lights_widths)
        self.lines_starts, _, self.lines_inverse = ragged_index(self.lines_widths)
        self.textures_starts, _, self.textures_inverse = ragged_index(self.textures_widths)
        assert self.lines_widths.sum() == self.lines_vals.shape[0]
        assert self.textures_widths.shape[0] == self.lines_vals.shape[0]
        assert self.textures_widths.sum() == n_tex
        assert self.lights_widths.sum() == self.lights_vals.shape[0]
        self.n_envs = len(self.lines_widths)
        self.struct = OrScenery(
            self.n_envs, self.n_agents, self.model.shape[0],
            _p(self.lights_vals), _p(self.lights_widths), _p(self.lights_starts),
            _p(self.lines_vals), _p(self.lines_widths), _p(self.lines_starts), _p(self.lines_inverse),
            _p(self.textures_vals), _p(self.textures_widths), _p(self.textures_starts), _p(self.textures_inverse),
            _p(self.model), _p(self.baked_vals),
            self.lines_vals.shape[0], self.lights_vals.shape[0], n_tex)


def _agents(agents):
    arrs = {k: _f32(agents[k]).copy() for k in ('angles', 'positions', 'angvelocity', 'velocity')}
    return arrs, OrAgents(*(_p(arrs[k]) for k in ('angles', 'positions', 'angvelocity', 'velocity')))


def bake(scene, cfg):
    """Returns the baked (sum T,) lighting and leaves it in ``scene.baked_vals`` (kernels.cu:270-293)."""
    assert isinstance(scene, Scene)
    rc = lib().oracle_bake(C.byref(scene.struct), C.byref(cfg))
    assert rc == 0
    return scene.baked_vals


def physics(scene, agents, cfg):
    """Returns (progress (N,A), new agents dict); inputs untouched (kernels.cu:212-230)."""
    assert isinstance(scene, Scene)
    arrs, st = _agents(agents)
    N, A = arrs['angles'].shape
    progress = np.zeros((N, A), np.float32)
    rc = lib().oracle_physics(C.byref(scene.struct), C.byref(st), _p(progress), C.byref(cfg))
    assert rc == 0
    return progress, arrs


def render(scene, agents, cfg):
    """Returns a dict of indices/locations/dots/distances/screen; rewrites the agent rows of
    ``scene.lines_vals`` like the reference does (kernels.cu:452-475)."""
    assert isinstance(scene, Scene)
    arrs, st = _agents(agents)
    N, A = arrs['angles'].shape
    R = cfg.res
    out = dict(
        indices=np.zeros((N, A, R), np.int32), locations=np.zeros((N, A, R), np.float32),
        dots=np.zeros((N, A, R), np.float32), distances=np.zeros((N, A, R), np.float32),
        screen=np.zeros((N, A, R, 3), np.float32))
    st_out = OrRender(*(_p(out[k]) for k in ('indices', 'locations', 'dots', 'distances', 'screen')))
    rc = lib().oracle_render(C.byref(scene.struct), C.byref(st), C.byref(st_out), C.byref(cfg))
    assert rc == 0
    return out
