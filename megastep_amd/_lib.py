"""Loader for ``libmegastep_hip.so`` - the hipcc-built gfx950 library behind ``include/megastep_hip.h``.

Counterpart of the reference's JIT build-and-load at import (reference: megastep/__init__.py:7-20): the library is
built in-tree with hipcc (``csrc/Makefile``) and bound with ctypes. There is NO fallback: if the library cannot be
built or loaded, or no HIP device is visible when a kernel is requested, this raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
# MEGASTEP_HIP_LIB points at an alternative build of the same ABI (A/B experiments); default is the in-tree build
LIB_PATH = os.environ.get('MEGASTEP_HIP_LIB') or os.path.join(CSRC, 'libmegastep_hip.so')
ABI_VERSION = 16

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int)


class MsConfig(C.Structure):
    _fields_ = [('agent_radius', C.c_float), ('res', C.c_int), ('fov', C.c_float), ('fps', C.c_float)]


class MsScenery(C.Structure):
    _fields_ = [
        ('n_envs', C.c_int), ('n_agents', C.c_int), ('n_model', C.c_int),
        ('lights_vals', C.c_void_p), ('lights_widths', C.c_void_p), ('lights_starts', C.c_void_p),
        ('lines_vals', C.c_void_p), ('lines_widths', C.c_void_p), ('lines_starts', C.c_void_p),
        ('lines_inverse', C.c_void_p),
        ('textures_vals', C.c_void_p), ('textures_widths', C.c_void_p), ('textures_starts', C.c_void_p),
        ('textures_inverse', C.c_void_p),
        ('model', C.c_void_p), ('baked_vals', C.c_void_p),
        ('n_lines_total', C.c_int), ('n_lights_total', C.c_int), ('n_texels_total', C.c_int),
        ('lg_vals', C.c_void_p), ('lg_starts', C.c_void_p), ('lg_geom', C.c_void_p), ('lg_cell', C.c_float),
        ('lg_max_cells', C.c_int), ('lg_list', C.c_void_p), ('lg_pool', C.c_void_p), ('lg_pool_size', C.c_int), ('lg_pool_rows', C.c_void_p),
        ('env_geom', C.c_void_p), ('bake_vis', C.c_void_p), ('bake_vis_starts', C.c_void_p), ('bake_vis_words', C.c_longlong),
        ('wg_cells', C.c_void_p), ('wg_starts', C.c_void_p), ('wg_geom', C.c_void_p), ('wg_cell', C.c_float),
        ('wg_reach_lo', C.c_float), ('wg_reach', C.c_float), ('wg_near', C.c_float), ('wg_pool', C.c_void_p), ('wg_pool_base', C.c_void_p), ('wg_near_rows', C.c_void_p),
        ('model_radius', C.c_float)]


class MsWallGridParent(C.Structure):
    _fields_ = [('cells', C.c_void_p), ('starts', C.c_void_p), ('geom', C.c_void_p), ('cell', C.c_float), ('pool', C.c_void_p)]


class MsAgents(C.Structure):
    _fields_ = [('angles', C.c_void_p), ('positions', C.c_void_p), ('angvelocity', C.c_void_p), ('velocity', C.c_void_p),
                ('headings', C.c_void_p)]


class MsMovement(C.Structure):
    _fields_ = [('actions', C.c_void_p), ('table', C.c_void_p), ('n_actions', C.c_int), ('keep', C.c_float)]


class MsStepExtras(C.Structure):
    _fields_ = [('respawn_mask', C.c_void_p), ('respawn_choice', C.c_void_p), ('spawn_positions', C.c_void_p),
                ('spawn_angles', C.c_void_p), ('n_spawns', C.c_int), ('respawn_after', C.c_int),
                ('lifespans', C.c_void_p), ('max_lifespans', C.c_void_p), ('fresh_max', C.c_void_p),
                ('imu', C.c_void_p), ('imu_ang_scale', C.c_float), ('imu_speed_scale', C.c_float)]


class MsRender(C.Structure):
    _fields_ = [('indices', C.c_void_p), ('locations', C.c_void_p), ('dots', C.c_void_p), ('distances', C.c_void_p),
                ('screen', C.c_void_p), ('workspace', C.c_void_p), ('obs_rgb', C.c_void_p), ('obs_depth', C.c_void_p),
                ('obs_subsample', C.c_int), ('obs_max_depth', C.c_float), ('obs_centre', C.c_void_p),
                ('seen_stamp', C.c_void_p), ('seen_epoch', C.c_void_p), ('seen_count', C.c_void_p)]


class MsDeathmatch(C.Structure):
    _fields_ = [('centre', C.c_void_p), ('positions', C.c_void_p), ('upper', C.c_void_p), ('clearance', C.c_float), ('hit_damage', C.c_float),
                ('tick_damage', C.c_float), ('health', C.c_void_p), ('damage', C.c_void_p), ('dead', C.c_void_p), ('reset_out', C.c_void_p),
                ('reward', C.c_void_p), ('health_obs', C.c_void_p), ('matchings', C.c_void_p)]


class MsExplorer(C.Structure):
    _fields_ = [('tally', C.c_void_p), ('before', C.c_void_p), ('lengths', C.c_void_p), ('epoch', C.c_void_p), ('over', C.c_void_p),
                ('slack', C.c_int), ('pixels', C.c_int), ('reset_out', C.c_void_p), ('reward', C.c_void_p), ('potential', C.c_void_p),
                ('length_out', C.c_void_p)]


#: every symbol include/megastep_hip.h (the boundary) and include/megastep_hip_test.h (test hooks) declare
SYMBOLS = ('ms_host_ray_interval_wide', 'ms_debug_ray_groups', 'ms_debug_last_render_groups', 'ms_debug_last_step_fused', 'ms_step_render', 'ms_move_step_render', 'ms_debug_ray_group_tail', 'ms_debug_physics_pack', 'ms_host_render_plan', 'ms_host_render_block', 'ms_host_physics_pack', 'ms_debug_pair_telemetry', 'ms_test_arithmetic', 'ms_abi_version', 'ms_strerror', 'ms_last_hip_error', 'ms_device_count', 'ms_bake', 'ms_physics', 'ms_move_physics',
           'ms_step_physics', 'ms_deathmatch_shoot', 'ms_explorer_books',
           'ms_render', 'ms_host_sincospi', 'ms_host_bake_point_bin', 'ms_host_bake_wall_bins',
           'ms_wallgrid_scan', 'ms_wallgrid_fill', 'ms_host_wall_hidden', 'ms_host_wall_sectors', 'ms_host_wallgrid_cell', 'ms_host_wall_arc',
           'ms_host_wedge_meets', 'ms_host_agents_apart', 'ms_host_wall_beyond_reach', 'ms_host_ray_interval', 'ms_host_fold_hits', 'ms_host_lightgrid_cell', 'ms_host_wall_reach')


def _source_hash():
    import hashlib
    h = hashlib.sha256()
    import glob
    # (the Makefile's SOURCES, in its order: the translation unit, the kernel files it includes, the two headers, itself)
    for path in (os.path.join(CSRC, 'megastep_hip.hip'), *sorted(glob.glob(os.path.join(CSRC, 'kernels', '*.h'))),
                 os.path.join(_HERE, '..', 'include', 'megastep_hip.h'), os.path.join(_HERE, '..', 'include', 'megastep_hip_test.h'),
                 os.path.join(CSRC, 'Makefile')):
        with open(path, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False):
    """Compiles csrc/megastep_hip.hip (and the csrc/kernels/*.h it includes) for gfx950 with hipcc (cross-compiles without a GPU). The library is stale when
    the hash of its sources differs from the one the Makefile recorded next to it (mtimes do not survive copies);
    concurrent callers (one rank per GPU) serialise on a lock file."""
    import fcntl
    stamp = LIB_PATH + '.srchash'
    want = _source_hash()

    def fresh():
        return os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == want

    if force or not fresh():
        with open(os.path.join(CSRC, '.build.lock'), 'w') as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if force or not fresh():                 # (someone else may have built it while we waited)
                import shutil
                hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
                if shutil.which('make') is None or shutil.which(hipcc) is None or not os.access(CSRC, os.W_OK):
                    raise NoToolchain(f'no make / {hipcc} on this host, or {CSRC} is read-only')
                proc = subprocess.run(['make', '-C', CSRC, '-B', 'libmegastep_hip.so'], capture_output=True, text=True)
                if proc.returncode != 0:
                    raise RuntimeError(f'hipcc build of libmegastep_hip.so failed:\n{proc.stdout}\n{proc.stderr}')
    return LIB_PATH


class NoToolchain(OSError):
    """The library cannot be rebuilt here for want of tools (not because its sources do not compile)."""


_lib = None


def lib():
    """The loaded library. (Re)builds the in-tree one first if it is missing or older than its sources; raises if that
    is impossible."""
    global _lib
    if _lib is None:
        if not os.environ.get('MEGASTEP_HIP_LIB'):
            try:
                build()                 # a no-op while the in-tree library matches its sources
            except NoToolchain as e:
                # no hipcc on this host, or a read-only tree: a library that is already there is loaded as it is and
                # the ABI check below decides; without one there is nothing to fall back to.  A compile ERROR is not
                # caught here: sources that do not build must never run a suite against yesterday's kernels.
                if not os.path.exists(LIB_PATH):
                    raise
                import warnings
                warnings.warn(f'{LIB_PATH} could not be rebuilt from the sources next to it ({e}); loading it as it is')
        handle = C.CDLL(LIB_PATH)
        missing = [s for s in SYMBOLS if not hasattr(handle, s)]
        if missing:
            raise ImportError(f'{LIB_PATH} does not export {missing}')
        handle.ms_abi_version.restype = C.c_int
        handle.ms_strerror.restype = C.c_char_p
        handle.ms_strerror.argtypes = [C.c_int]
        handle.ms_last_hip_error.restype = C.c_int
        handle.ms_device_count.restype = C.c_int
        handle.ms_bake.argtypes = [C.POINTER(MsScenery), C.POINTER(MsConfig), C.c_void_p]
        handle.ms_physics.argtypes = [C.POINTER(MsScenery), C.POINTER(MsAgents), C.c_void_p, C.POINTER(MsConfig), C.c_void_p]
        handle.ms_move_physics.argtypes = [C.POINTER(MsScenery), C.POINTER(MsAgents), C.POINTER(MsMovement), C.c_void_p,
                                           C.POINTER(MsConfig), C.c_void_p]
        handle.ms_step_physics.argtypes = [C.POINTER(MsScenery), C.POINTER(MsAgents), C.POINTER(MsMovement), C.POINTER(MsStepExtras),
                                           C.c_void_p, C.POINTER(MsConfig), C.c_void_p]
        handle.ms_step_render.argtypes = [C.POINTER(MsScenery), C.POINTER(MsAgents), C.c_void_p, C.POINTER(MsRender), C.POINTER(MsConfig), C.c_void_p]
        handle.ms_step_render.restype = C.c_int
        handle.ms_move_step_render.argtypes = [C.POINTER(MsScenery), C.POINTER(MsAgents), C.POINTER(MsMovement), C.POINTER(MsStepExtras),
                                               C.c_void_p, C.POINTER(MsRender), C.POINTER(MsConfig), C.c_void_p]
        handle.ms_move_step_render.restype = C.c_int
        handle.ms_debug_last_step_fused.argtypes = []
        handle.ms_debug_last_step_fused.restype = C.c_int
        handle.ms_explorer_books.argtypes = [C.c_int, C.POINTER(MsExplorer), C.c_void_p]
        handle.ms_explorer_books.restype = C.c_int
        handle.ms_deathmatch_shoot.argtypes = [C.c_int, C.c_int, C.POINTER(MsDeathmatch), C.c_void_p]
        handle.ms_deathmatch_shoot.restype = C.c_int
        handle.ms_render.argtypes = [C.POINTER(MsScenery), C.POINTER(MsAgents), C.POINTER(MsRender), C.POINTER(MsConfig), C.c_void_p]
        handle.ms_host_sincospi.argtypes = [C.c_float, _f32p, _f32p]
        handle.ms_host_bake_point_bin.argtypes = [C.c_float]*4
        handle.ms_host_bake_point_bin.restype = C.c_int
        handle.ms_host_bake_wall_bins.argtypes = [C.c_float]*6 + [_i32p, _i32p]
        handle.ms_host_bake_wall_bins.restype = None
        handle.ms_wallgrid_scan.argtypes = [C.POINTER(MsScenery), C.POINTER(MsWallGridParent), C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
        handle.ms_wallgrid_fill.argtypes = [C.POINTER(MsScenery), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
        handle.ms_host_wall_sectors.argtypes = [C.c_float]*4 + [_f32p, _i32p, _i32p, _f32p, _i32p]
        handle.ms_host_wall_sectors.restype = None
        handle.ms_host_wall_arc.argtypes = [C.c_float]*4 + [_f32p, _i32p, _i32p]
        handle.ms_host_wall_arc.restype = None
        handle.ms_host_wedge_meets.argtypes = [C.c_float]*4 + [C.c_int, C.c_int]
        handle.ms_host_wedge_meets.restype = C.c_int
        handle.ms_host_ray_interval.argtypes = [_f32p, _f32p, C.c_int, C.c_float, C.c_float, C.c_int, _i32p, _i32p]
        handle.ms_host_ray_interval.restype = None
        handle.ms_host_ray_interval_wide.argtypes = [_f32p, _f32p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, _i32p, _i32p]
        handle.ms_host_ray_interval_wide.restype = None
        handle.ms_debug_ray_groups.argtypes = [C.c_int]
        handle.ms_debug_ray_groups.restype = C.c_int
        handle.ms_debug_last_render_groups.argtypes = []
        handle.ms_debug_last_render_groups.restype = C.c_int
        handle.ms_debug_ray_group_tail.argtypes = [C.c_float, C.c_int]
        handle.ms_debug_physics_pack.argtypes = [C.c_int]
        handle.ms_debug_physics_pack.restype = C.c_int
        handle.ms_host_render_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.POINTER(C.c_int)]
        handle.ms_host_render_plan.restype = C.c_longlong
        handle.ms_host_render_block.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_longlong, C.POINTER(C.c_int)]
        handle.ms_host_render_block.restype = C.c_int
        handle.ms_host_physics_pack.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        handle.ms_host_physics_pack.restype = C.c_int
        handle.ms_debug_ray_group_tail.restype = C.c_int
        handle.ms_debug_pair_telemetry.argtypes = [C.c_int]
        handle.ms_debug_pair_telemetry.restype = C.c_int
        handle.ms_test_arithmetic.argtypes = [C.c_void_p]*7 + [C.c_longlong, C.c_void_p]
        handle.ms_test_arithmetic.restype = C.c_int
        handle.ms_host_fold_hits.argtypes = [_f32p, _i32p, C.c_int, _i32p, _f32p, _i32p]
        handle.ms_host_fold_hits.restype = C.c_int
        handle.ms_host_lightgrid_cell.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float,
                                                  C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        handle.ms_host_lightgrid_cell.restype = C.c_int
        handle.ms_host_agents_apart.argtypes = [_f32p, _f32p, C.c_float]
        handle.ms_host_agents_apart.restype = C.c_int
        handle.ms_host_wall_beyond_reach.argtypes = [_f32p, _f32p, C.c_float]
        handle.ms_host_wall_beyond_reach.restype = C.c_int
        handle.ms_host_wall_reach.argtypes = [_f32p, C.c_float]
        handle.ms_host_wall_reach.restype = C.c_float
        handle.ms_host_wall_hidden.argtypes = [C.c_float]*4 + [_f32p, _f32p, C.c_float]
        handle.ms_host_wall_hidden.restype = C.c_int
        handle.ms_host_wallgrid_cell.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int,
                                                 C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        handle.ms_host_wallgrid_cell.restype = None
        for name in ('ms_bake', 'ms_physics', 'ms_move_physics', 'ms_step_physics', 'ms_render', 'ms_wallgrid_scan', 'ms_wallgrid_fill'):
            getattr(handle, name).restype = C.c_int
        if handle.ms_abi_version() != ABI_VERSION:
            raise ImportError(f'{LIB_PATH} has ABI {handle.ms_abi_version()}, this package needs {ABI_VERSION}; rebuild it')
        _lib = handle
    return _lib


def check(code):
    """Turns a negative MS_E* return into a RuntimeError, the way the reference's AT_ASSERTs surface in Python."""
    if code != 0:
        h = lib()
        raise RuntimeError(f'megastep_hip: {h.ms_strerror(code).decode()} (code {code}, hipError {h.ms_last_hip_error()})')
