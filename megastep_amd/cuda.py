"""The ``megastep.cuda`` operator surface on MI355X.

Mirrors the reference's pybind module ``megastepcuda`` (reference: megastep/src/wrappers.cpp:30-173) name for name -
``initialize, bake, physics, render, Ragged1D/2D/3D, Agents, Scenery, Render, Physics`` - but every compute entry
point is a hand-written gfx950 kernel reached through the C-ABI in ``include/megastep_hip.h``. Tensors stay torch-owned
(PyTorch-ROCm is the allocator/stream plumbing); the kernels run on ``torch.cuda.current_stream()``.

There is no CPU implementation here, exactly as in the reference ("If you haven't got CUDA, megastep will not work",
reference: docs/faq.rst:23-26): calling bake/physics/render on non-GPU tensors raises.
"""
import ctypes as C
import numbers
import os
import sys
import torch
from . import _lib

# ---------------------------------------------------------------------------------------------------------------------
# initialize                                                                   reference: kernels.cu:18-27
# ---------------------------------------------------------------------------------------------------------------------
_config = None


_switched = __import__('threading').local()


def _ab_switches():
    """A/B switches of the library that tools set through the environment (the library itself reads none on its hot path).
    The library's pins are per THREAD (megastep_hip.hip: thread_local), so they are applied once on every thread that launches -
    physics() and render() call this - not only on the one that called initialize() (an A/B run that stepped from a
    worker thread compared two identical builds)."""
    if getattr(_switched, 'done', False):
        return
    _switched.done = True
    g = os.environ.get('MEGASTEP_RAY_GROUPS')
    if g:
        _lib.lib().ms_debug_ray_groups(int(g))           # 64-ray groups per render wave: 1, 2, 4 (default: by resolution)
    k = os.environ.get('MEGASTEP_PHYSICS_PACK')
    if k:
        _lib.lib().ms_debug_physics_pack(int(k))         # envs a physics wave takes side by side (default: by the world's size)
    t, e = os.environ.get('MEGASTEP_RAY_GROUP_TAIL'), os.environ.get('MEGASTEP_RAY_GROUP_TAIL_ENVS')
    if t or e:                                           # the one-group waves at the end of a launch of wide ones: in rounds / in envs
        _lib.lib().ms_debug_ray_group_tail(float(t) if t else -1., int(e) if e else -1)


def config(agent_radius, res, fov, fps):
    """The four constants of :func:`initialize` as a value (the C-ABI's ``MsConfig``, passed by value with every launch):
    what a :class:`~megastep_amd.core.Core` keeps for itself and hangs on its ``Agents``, so that several Cores of
    different resolutions, fields of view or frame rates live side by side in one process - on one device or several."""
    if not (0 < fov < 180):
        raise RuntimeError('fov must be in (0, 180) degrees')
    if res <= 0 or fps <= 0 or agent_radius <= 0:
        raise RuntimeError('agent_radius, res and fps must be positive')
    return _lib.MsConfig(float(agent_radius), int(res), float(fov), float(fps))


def initialize(agent_radius, res, fov, fps):
    """Sets the constants used by :func:`physics` and :func:`render` for callers that hand them bare tensors' holders
    (reference: wrappers.cpp:53) - the drop-in path. As in the reference this is process-global; unlike it, nothing
    device-side is mutated (the values travel by value with every launch), and it is only the FALLBACK: a call that is
    given ``config=``, or whose ``agents`` carry one (every ``Core``'s do), never looks at it."""
    global _config
    _config = config(agent_radius, res, fov, fps)
    _ab_switches()


def _cfg(agents=None, explicit=None):
    """The constants of one call: the ``config=`` argument, else the ones the agents' Core hung on them, else initialize()'s."""
    if explicit is not None:
        if not isinstance(explicit, _lib.MsConfig):
            raise RuntimeError('config must come from megastep_amd.cuda.config(agent_radius, res, fov, fps)')
        return explicit
    own = getattr(agents, '_config', None)
    if own is not None:
        return own
    if _config is None:
        raise RuntimeError('megastep_amd.cuda.initialize(agent_radius, res, fov, fps) has not been called')
    return _config


# ---------------------------------------------------------------------------------------------------------------------
# checks                                                                       reference: common.h:12-14,33-37
# ---------------------------------------------------------------------------------------------------------------------
def _check(t, name, dtype, ndim):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f'{name} must be a tensor')
    if not t.is_contiguous():
        raise RuntimeError(f'{name} must be contiguous')
    if t.dtype != dtype:
        raise RuntimeError(f'{name} must have dtype {dtype}, not {t.dtype}')
    if t.ndim != ndim:
        raise RuntimeError(f'{name} must be {ndim}-dimensional, not {t.ndim}')
    return t


def _require_gpu(*tensors):
    dev = tensors[0].device
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError('megastep_amd kernels need GPU (HIP) tensors; got a tensor on ' + str(t.device))
        if t.device != dev:
            raise RuntimeError(f'all tensors must live on one device; got {t.device} and {dev}')
    return dev


# ---------------------------------------------------------------------------------------------------------------------
# Ragged                                                                       reference: common.h:91-155
# ---------------------------------------------------------------------------------------------------------------------
class _Ragged:
    """Arrays-of-arrays over one contiguous backing tensor (reference: common.h:102-155, wrappers.cpp:14-28).

    ``starts/ends/inverse`` are int32 like ``widths``. Zero-width rows are legal here (the reference's scatter-based
    ``inverses`` mis-handles them, common.h:91-98): ``inverse`` simply never mentions them."""
    _ndim = None

    def __init__(self, vals, widths):
        _check(vals, 'vals', torch.float32, self._ndim)
        _check(widths, 'widths', torch.int32, 1)
        if vals.device != widths.device:
            raise RuntimeError('vals and widths must be on the same device')
        ends = widths.cumsum(0).to(torch.int32)
        total = int(ends[-1]) if len(widths) else 0
        if total != vals.shape[0]:
            raise RuntimeError(f'widths sum to {total} but vals has {vals.shape[0]} rows')
        self._vals, self._widths = vals, widths
        self._starts, self._ends = ends - widths, ends
        self._inverse = torch.repeat_interleave(
            torch.arange(len(widths), dtype=torch.int32, device=widths.device), widths.long())

    vals = property(lambda self: self._vals)
    widths = property(lambda self: self._widths)
    starts = property(lambda self: self._starts)
    ends = property(lambda self: self._ends)
    inverse = property(lambda self: self._inverse)

    def __len__(self):
        return len(self._widths)

    def __getitem__(self, x):
        if isinstance(x, numbers.Integral):
            return self._vals[int(self._starts[x]):int(self._ends[x])]
        if isinstance(x, slice):
            start, stop, step = x.indices(len(self._widths))
            if step != 1:
                raise IndexError('Ragged slices must have step 1')
            if stop <= start:
                return type(self)(self._vals[:0], self._widths[:0])
            return type(self)(self._vals[int(self._starts[start]):int(self._ends[stop - 1])], self._widths[start:stop])
        raise TypeError(f"Can't index a Ragged with a {type(x).__name__}")

    def size(self, i):
        return self._vals.shape[i]

    def clone(self):
        return type(self)(self._vals.clone(), self._widths.clone())

    def numpyify(self):
        from .ragged import RaggedNumpy
        return RaggedNumpy(self._vals.detach().cpu().numpy().copy(), self._widths.detach().cpu().numpy().copy())

    def to(self, device):
        return type(self)(self._vals.to(device), self._widths.to(device))

    def __repr__(self):
        return f'{type(self).__name__}(vals={tuple(self._vals.shape)}, widths=({len(self._widths)},))'


class Ragged1D(_Ragged):
    _ndim = 1


class Ragged2D(_Ragged):
    _ndim = 2


class Ragged3D(_Ragged):
    _ndim = 3


# ---------------------------------------------------------------------------------------------------------------------
# Agents / Scenery / Render / Physics                                          reference: common.h:157-226
# ---------------------------------------------------------------------------------------------------------------------
class Agents:
    """Holds the (N, A[, 2]) state tensors of the agents (reference: common.h:157-177, wrappers.cpp:103-120).
    :func:`physics` updates them in place."""

    #: ``False`` sends ms_render through its own heading kernel instead of the cache ms_physics leaves (tests, A/B runs)
    HEADING_CACHE = True

    def __init__(self, angles, positions, angvelocity, velocity, config=None):
        #: the constants :func:`physics` / :func:`render` use for these agents (:func:`config`); None: initialize()'s
        self._config = config
        self._angles = _check(angles, 'angles', torch.float32, 2)
        self._positions = _check(positions, 'positions', torch.float32, 3)
        self._angvelocity = _check(angvelocity, 'angvelocity', torch.float32, 2)
        self._velocity = _check(velocity, 'velocity', torch.float32, 3)
        n, a = angles.shape
        if positions.shape != (n, a, 2) or velocity.shape != (n, a, 2) or angvelocity.shape != (n, a):
            raise RuntimeError('agent tensors must be (N, A), (N, A, 2), (N, A), (N, A, 2)')
        # The heading cache (include/megastep_hip.h, MsAgents.headings): physics leaves each agent's sin/cos there for
        # the next render. `_struct` carries it; `_plain` does not, for renders before any physics call has filled it.
        self._headings = torch.full((n, a, 4), float('nan'), dtype=torch.float32, device=angles.device)
        ptrs = (angles.data_ptr(), positions.data_ptr(), angvelocity.data_ptr(), velocity.data_ptr())
        self._struct = _lib.MsAgents(*ptrs, self._headings.data_ptr())
        self._plain = _lib.MsAgents(*ptrs, None)
        self._cached = False
        self._epoch = 0             # moved on by every physics call and every respawn (modules.IMU's stale-reading check)
        # (measured at 16 k to 262 k agents: one launch and one kernel boundary fewer per step, 1-3 % of the step)
        self._use_cache = bool(self.HEADING_CACHE)
        devices = {t.device for t in (angles, positions, angvelocity, velocity)}
        self._dev = devices.pop() if len(devices) == 1 else None         # None: tensors on mixed devices

    angles = property(lambda self: self._angles)
    positions = property(lambda self: self._positions)
    angvelocity = property(lambda self: self._angvelocity)
    velocity = property(lambda self: self._velocity)

    def state(self, e):
        from . import arrdict
        return arrdict.arrdict(angles=self._angles[e], positions=self._positions[e],
                               angvelocity=self._angvelocity[e], velocity=self._velocity[e])


class Scenery:
    """Holds the geometry, lights and textures of every env (reference: common.h:179-214, wrappers.cpp:122-145).

    ``lines`` is ragged per env with the ``n_agents*len(model)`` agent lines first; ``textures`` and ``baked`` are
    ragged per *line*. ``baked`` starts as ones and is filled in by :func:`bake`."""

    def __init__(self, n_agents, lights, lines, textures, model, geom=None):
        if not isinstance(lights, Ragged2D) or not isinstance(lines, Ragged3D) or not isinstance(textures, Ragged2D):
            raise RuntimeError('lights, lines, textures must be Ragged2D, Ragged3D, Ragged2D')
        _check(model, 'model', torch.float32, 3)
        if lights.vals.shape[1:] != (3,) or lines.vals.shape[1:] != (2, 2) or textures.vals.shape[1:] != (3,) \
                or model.shape[1:] != (2, 2):
            raise RuntimeError('lights/lines/textures/model rows must be (3,), (2, 2), (3,), (2, 2)')
        if len(lights) != len(lines):
            raise RuntimeError('lights and lines must have one row per env')
        if len(textures) != lines.vals.shape[0]:
            raise RuntimeError('textures must have one row per line')
        self._n_agents = int(n_agents)
        self._lights, self._lines, self._textures, self._model = lights, lines, textures, model
        self._baked = Ragged1D(torch.ones_like(textures.vals[:, 0]).contiguous(), textures.widths)
        # how far the agent's outline reaches from its origin (MsScenery.model_radius): inside the near plane, as the
        # reference's is, an agent's rays cannot hit its own outline and the renderer does not try
        radius = float(model.reshape(-1, 2).norm(dim=1).max()) if model.numel() else 0.
        self._model_radius = radius if radius == radius and radius < float('inf') else 0.
        # Beyond the reference: `geom` (n_envs,) int32 names, for every env, the first env with bit-identical walls
        # and light positions (see MsScenery.env_geom). bake() then does the expensive part once per distinct
        # floorplan, and such envs share one light grid. None: every env stands alone.
        if geom is not None:
            _check(geom, 'geom', torch.int32, 1)
            if len(geom) != len(lines) or geom.device != lines.vals.device:
                raise RuntimeError('geom must have one entry per env, on the same device as the lines')
        self._geom = geom
        self._struct = None
        self._lg = None             # the light grid's tensors; made by _as_struct unless sharding carried one over
        self._wg_sum = None         # checksum of the static walls the wall grid was built from
        self._wg_weights = None
        self._wg = None             # the wall grid's tensors (cells, starts, geom, cell, reaches, near plane, pool, near rows, pool bases); made by bake()
        self._wg_report = self._lg_report = None      # what was built, at which cell size, in how many bytes (grid_report())
        self._dev = None

    n_agents = property(lambda self: self._n_agents)
    lights = property(lambda self: self._lights)
    lines = property(lambda self: self._lines)
    textures = property(lambda self: self._textures)
    model = property(lambda self: self._model)
    baked = property(lambda self: self._baked)
    geom = property(lambda self: self._geom)

    def state(self, e):
        from . import dotdict
        s, t = int(self._lines.starts[e]), int(self._lines.ends[e])
        return dotdict.dotdict(n_agents=self._n_agents, lights=self._lights[e], lines=self._lines[e],
                               textures=self._textures[s:t], model=self._model, baked=self._baked[s:t])

    LIGHT_GRID = True           # False: no light grid (ms_render's two-kernel path; tests)
    #: cell size of the light grid, metres. (0.25 -> 0.125 in round 4: half as many rays that land on an agent still have a
    #: light the grid cannot call - the rays whose waves a launch ends up waiting for; four times the cells.)
    LIGHT_GRID_CELL = float(os.environ.get('MEGASTEP_LIGHT_GRID_CELL', .125))     # (the environment switch is for A/B runs)
    LIGHT_GRID_POOL = 12        # pool words per cell (4 bytes each) for the candidate lists
    #: what the light grid may take, bytes. A cell costs 16 B of verdicts + 8 B of list header + LIGHT_GRID_POOL x (4 B candidate
    #: + 16 B of its wall's row) = 264 B: at 0.125 m cells 17 KB per square metre, 3-4 MB per benchmark floorplan, 14 MB per large
    #: one - 1024 plans (the headline world) 3.4 GB, 4096 large ones 54 GB. Over budget the grid first goes without the
    #: candidates' rows (72 B a cell: ms_render then fetches a candidate's wall one trip later, the round-3 behaviour), then
    #: doubles its cells (a quarter of them, and more lights left UNKNOWN per cell: slower for the rays that land on an agent,
    #: the same bits) until it fits. `Scenery.grid_report()` says what was built.
    #: None: an eighth of the device's memory (36 GB of an MI355X's 288; 8 GiB for tensors that are not on a GPU).
    LIGHT_GRID_BYTES = int(float(os.environ['MEGASTEP_LIGHT_GRID_BYTES'])) if os.environ.get('MEGASTEP_LIGHT_GRID_BYTES') else None

    def _light_grid_budget(self):
        if self.LIGHT_GRID_BYTES is not None:
            return int(self.LIGHT_GRID_BYTES)
        dev = self._lines.vals.device
        return _memory_share(dev, 8) if dev.type == 'cuda' else 8 << 30

    def _light_grid(self):
        """Storage and geometry of the light grid (see include/megastep_hip.h): a uniform grid over each env's walls,
        half a metre of slack around them: per cell the lights' verdicts and a candidate list drawn from a shared pool.
        `bake` fills it in; zeros mean 'unknown, test every wall', which is always safe. Envs that share their
        geometry (`geom`) share their representative's cells. 264 bytes per cell (16 verdicts + 8 list header + 12 pool
        words of 4 + 16 each), i.e. hundreds of times the floorplan's own lines - hence the sharing, the byte budget
        (LIGHT_GRID_BYTES) and no grid at all for single-agent sceneries, whose rays never land on an agent."""
        ln = self._lines
        dev = ln.vals.device
        n_envs = len(ln)
        lo, hi = self._wall_bounds()
        origin = torch.floor(lo) - .5
        rep = torch.arange(n_envs, device=dev) if self._geom is None else self._geom.long()
        is_rep = rep == torch.arange(n_envs, device=dev)
        cell, with_rows = float(self.LIGHT_GRID_CELL), True
        budget = self._light_grid_budget()
        while True:
            dims = torch.ceil((hi + .5 - origin)/cell).clamp(1, 4096)
            # the grid holds 64 lights per env: an env with more gets no cells, and the renderer meets every wall for the
            # rays that land on an agent there (the other envs keep their grids, and the launch stays one kernel)
            dims = torch.where((self._lights.widths <= 64)[:, None], dims, torch.zeros_like(dims))
            cells = (dims[:, 0]*dims[:, 1]).long()
            own = cells*is_rep                                             # members own no cells
            total = int(own.sum())
            words = min(1 + self.LIGHT_GRID_POOL*total, 2**31 - 1)
            size = 24*(total + 1) + (20 if with_rows else 4)*words
            if size <= budget or cell >= 8.:
                break
            if with_rows:
                with_rows = False
            else:
                cell *= 2
        starts = (own.cumsum(0) - own)[rep].to(torch.int32)
        geom = torch.cat([origin, dims], 1).float()[rep].contiguous()
        vals = torch.zeros((total + 1, 4), dtype=torch.int32, device=dev)     # (+ a row for rays outside the last env's grid to read)
        lists = torch.zeros((total + 1, 2), dtype=torch.int32, device=dev)
        pool = torch.zeros(words, dtype=torch.int32, device=dev)
        # the candidates' walls, next to their entries (MsScenery.lg_pool_rows; optional)
        rows = torch.zeros((words, 4), dtype=torch.float32, device=dev) if with_rows else None
        self._lg_report = dict(bytes=size, cell=cell, cells=total, candidate_rows=with_rows, budget=budget,
                               floorplans=int(is_rep.sum()))
        if os.environ.get('MEGASTEP_VERBOSE'):
            print(f'megastep_amd: light grid of {size/2**20:.0f} MiB: {total} cells of {cell:g} m over {int(is_rep.sum())} floorplans'
                  + ('' if with_rows else ', without candidate rows') + (f' (asked for {self.LIGHT_GRID_CELL:g} m: over the '
                  f'{budget/2**30:.1f} GiB budget)' if cell != self.LIGHT_GRID_CELL or not with_rows else ''), flush=True)
        return vals, starts.contiguous(), geom, cell, max(int(cells.max()), 1), lists, pool, rows

    def grid_report(self):
        """What bake() built around the floorplans, for logs and bench lines: {'wall_grid': {bytes, cell, floorplans, ...} or
        None, 'light_grid': {bytes, cell, cells, candidate_rows, ...} or None, 'bake_seconds': {lighting, wall_grid} of the last
        cuda.bake()} - the sizes actually allocated and the cell sizes actually used (either grid coarsens itself to stay inside
        its byte budget)."""
        marks = getattr(self, '_bake_marks', None)
        if getattr(self, '_bake_s', None) is None and marks is not None and marks[0] is not None:
            ev, wall_grid = marks
            ev[2].synchronize()
            self._bake_s = dict(lighting=ev[0].elapsed_time(ev[1])*1e-3, wall_grid=ev[1].elapsed_time(ev[2])*1e-3 if wall_grid else 0.)
        return dict(wall_grid=self._wg_report, light_grid=self._lg_report, bake_seconds=getattr(self, '_bake_s', None))

    def _wall_bounds(self):
        """(n_envs, 2) lower and upper corner of each env's static walls (finite coordinates only; 0, 0 without any)."""
        ln = self._lines
        dev = ln.vals.device
        n_envs = len(ln)
        env = ln.inverse.long()
        static = (torch.arange(ln.vals.shape[0], device=dev) - ln.starts.long()[env]) >= self._n_agents*self._model.shape[0]
        env, pts = env[static], ln.vals[static]              # the walls; agent rows move
        big = torch.finfo(torch.float32).max
        fin = torch.isfinite(pts)
        idx = env[:, None].expand(-1, 2)
        lo = torch.full((n_envs, 2), big, device=dev).scatter_reduce_(0, idx, torch.where(fin, pts, big).amin(1), 'amin')
        hi = torch.full((n_envs, 2), -big, device=dev).scatter_reduce_(0, idx, torch.where(fin, pts, -big).amax(1), 'amax')
        empty = (lo > hi).any(1)                             # an env without (finite) walls gets a 1 x 1 grid at the origin
        lo[empty], hi[empty] = 0., 0.
        return lo, hi

    def _as_struct(self):
        if self._struct is None:
            li, ln, tx = self._lights, self._lines, self._textures
            if self._lg is None:
                # (one agent per env: no ray ever lands on an agent line, so nothing would consult the grid)
                wanted = self._n_agents > 1 and self.LIGHT_GRID
                self._lg = self._light_grid() if wanted else (None, None, None, 0., 0, None, None, None)
            lg = self._lg
            self._struct = _lib.MsScenery(
                len(ln), self._n_agents, self._model.shape[0],
                li.vals.data_ptr(), li.widths.data_ptr(), li.starts.data_ptr(),
                ln.vals.data_ptr(), ln.widths.data_ptr(), ln.starts.data_ptr(), ln.inverse.data_ptr(),
                tx.vals.data_ptr(), tx.widths.data_ptr(), tx.starts.data_ptr(), tx.inverse.data_ptr(),
                self._model.data_ptr(), self._baked.vals.data_ptr(),
                ln.vals.shape[0], li.vals.shape[0], tx.vals.shape[0],
                *(t.data_ptr() if t is not None else None for t in lg[:3]), lg[3], lg[4],
                *(t.data_ptr() if t is not None else None for t in lg[5:7]), lg[6].shape[0] if lg[6] is not None else 0,
                lg[7].data_ptr() if len(lg) > 7 and lg[7] is not None else None,
                self._geom.data_ptr() if self._geom is not None else None, None, None, 0,
                *self._wall_grid_fields(), self._model_radius)
        return self._struct

    def _wall_grid_fields(self):
        wg = self._wg
        if wg is None:
            return (None, None, None, 0., 0., 0., 0., None, None, None)
        cells, starts, geom, cell, reach_lo, reach, near, pool, rows, pool_base = wg
        return (cells.data_ptr(), starts.data_ptr(), geom.data_ptr(), cell, reach_lo, reach, near, pool.data_ptr(), pool_base.data_ptr(), rows.data_ptr())

    #: the wall grid (include/megastep_hip.h, MsScenery.wg_*): cell size in metres; the step lengths its collision lists
    #: cover (0.7 m: the momentum module at its defaults never reaches farther; 1.3 m: nor does the simple one at 10 fps);
    #: the near plane its visibility lists allow for; how much memory it may take - beyond that the cells are doubled in
    #: size, twice at most, then it is left out
    WALL_GRID = os.environ.get('MEGASTEP_WALL_GRID', '1') != '0'       # (the environment switch is for A/B runs)
    WALL_GRID_CELL = float(os.environ.get('MEGASTEP_WALL_GRID_CELL', .25))        # (the environment switch is for A/B runs)
    WALL_GRID_REACH = (.7, 1.3)
    WALL_GRID_NEAR = .12
    #: (bytes; None: a quarter of the device's memory - 72 GB of an MI355X's 288: the headline world's 1024 plans take 1.7 GB,
    #: C2's 4096 plans 6.9 GB, 4096 large plans - C5's share on the reference's floorplan diversity - 35 GB at 0.25 m cells;
    #: with rounds 3-4's flat 8 GiB that world fell back to 1 m cells and a step a third slower)
    WALL_GRID_BYTES = int(float(os.environ['MEGASTEP_WALL_GRID_BYTES'])) if os.environ.get('MEGASTEP_WALL_GRID_BYTES') else None
    WALL_GRID_MAX_CELLS = 1 << 18       # per floorplan
    WALL_GRID_SCRATCH = 1 << 30         # bytes of bitmaps in flight while it is built
    WALL_GRID_COARSE = 4                # the parent level's cells, in cells (None: one level, every wall a candidate)

    def _wall_grid_budget(self):
        if self.WALL_GRID_BYTES is not None:
            return int(self.WALL_GRID_BYTES)
        return _memory_share(self._device(), 4)

    def _scan_level(self, cell, parent, final, usable):
        """One level of the wall grid: cells of size `cell` over every representative floorplan, scanned (against the
        parent level's lists, if there is one) and filled. Returns (hdr (cells + 1, 4) int32 holding uint32s, starts, geom,
        pool int16, near rows float32 or None, bytes) - or None where there is nothing to build / the budget is exceeded."""
        dev = self._device()
        ln = self._lines
        n_envs = len(ln)
        af = self._n_agents*self._model.shape[0]
        walls = (ln.widths.long() - af).clamp(min=0)
        arange = torch.arange(n_envs, device=dev)
        rep = arange if self._geom is None else self._geom.long()
        is_rep = rep == arange
        lo, hi = self._wall_bounds()
        h = _lib.lib()
        origin = torch.floor(lo) - .5
        dims = torch.ceil((hi + .5 - origin)/self.WALL_GRID_CELL).clamp(min=1)      # in the finest cells ...
        scale = round(cell/self.WALL_GRID_CELL)
        dims = torch.ceil(dims/scale)                                               # ... so that every level covers the same ground
        cells = (dims[:, 0]*dims[:, 1])
        dims = torch.where(usable[:, None], dims, torch.zeros_like(dims))           # (the same envs at every level)
        cells = torch.where(usable, cells, torch.zeros_like(cells)).long()
        own = cells*is_rep                                             # members own no cells
        starts = (own.cumsum(0) - own)[rep].to(torch.int32).contiguous()
        geom = torch.cat([origin, dims], 1).float()[rep].contiguous()
        total = int(own.sum())
        if total == 0:
            return None
        w32 = (walls + 31)//32
        row_words = 3*own*w32                                          # bitmap words per representative
        reps = torch.nonzero(own > 0).flatten()
        struct = _lib.MsScenery.from_buffer_copy(self._as_struct())
        struct.wg_starts, struct.wg_geom, struct.wg_cell = starts.data_ptr(), geom.data_ptr(), cell
        struct.wg_reach_lo, struct.wg_reach, struct.wg_near = *self.WALL_GRID_REACH, self.WALL_GRID_NEAR
        par = None
        if parent is not None:
            par = _lib.MsWallGridParent(parent[0].data_ptr(), parent[1].data_ptr(), parent[2].data_ptr(), parent[3], parent[4].data_ptr())
        counts = torch.zeros((total, 3), dtype=torch.int32, device=dev)
        # groups of representatives whose bitmaps fit the scratch budget (and a launch's grid)
        words = row_words[reps]
        group = torch.maximum((words.cumsum(0) - words)*4//self.WALL_GRID_SCRATCH, torch.arange(len(reps), device=dev)//60000)
        bounds = [0] + (torch.nonzero(group[1:] != group[:-1]).flatten() + 1).tolist() + [len(reps)]
        pools, nears = [], []
        budget = self._wall_grid_budget()
        cell_rows = torch.zeros((total + 1, 4), dtype=torch.int64, device=dev)     # (+ the row agents outside the last grid read)
        # where each floorplan's vis lists start in the pool (MsScenery.wg_pool_base; the final level's cells count their
        # "first vis entry" from there: a world of thousands of large plans has more entries than 32 bits number)
        pool_base = torch.zeros(n_envs, dtype=torch.int64, device=dev)
        struct.wg_pool_base = pool_base.data_ptr()
        base_p = base_n = 0
        with _on(dev):
            for g0, g1 in zip(bounds[:-1], bounds[1:]):
                r = reps[g0:g1]
                r32 = r.to(torch.int32).contiguous()
                gw = row_words[r]
                bits_starts = torch.zeros(n_envs, dtype=torch.int64, device=dev)
                bits_starts[r] = gw.cumsum(0) - gw
                bits = torch.zeros(max(int(gw.sum()), 1), dtype=torch.int32, device=dev)
                mc = int(cells[r].max())
                groups = int(parent[5][r].max()) if parent is not None else (mc + 3)//4
                _lib.check(h.ms_wallgrid_scan(C.byref(struct), C.byref(par) if par is not None else None, r32.data_ptr(), len(r), groups,
                                              bits_starts.data_ptr(), bits.data_ptr(), counts.data_ptr(), _stream(dev)))
                # the cells of this group, in storage order, and their lists' places in the pools
                span = own[r]
                rows = torch.repeat_interleave(starts[r].long() - (span.cumsum(0) - span), span) + torch.arange(int(span.sum()), device=dev)
                cnt = counts[rows].long()                              # (cells, 3): vis, near within the short reach, near beyond
                n_vis, n_near = cnt[:, 0], cnt[:, 1] + cnt[:, 2]
                if final:                                              # vis -> pool (entries), near -> rows
                    off_v, off_n = n_vis.cumsum(0) - n_vis, n_near.cumsum(0) - n_near + base_n
                    add_p, add_n = int(n_vis.sum()), int(n_near.sum())
                    # ... the vis lists numbered from their floorplan's own start
                    first_cell = span.cumsum(0) - span                 # each representative's first cell among this group's
                    rep_off = off_v[first_cell.clamp(max=len(off_v) - 1)]
                    pool_base[r] = rep_off + base_p
                    off_v = off_v - torch.repeat_interleave(rep_off, span)
                else:                                                  # both lists as indices, back to back
                    both = torch.stack([n_vis, n_near], 1).flatten()
                    off = (both.cumsum(0) - both + base_p).view(-1, 2)
                    off_v, off_n = off[:, 0], off[:, 1]
                    add_p, add_n = int(both.sum()), 0
                if (4 if final else 2)*(base_p + add_p) + 16*(base_n + add_n) > budget \
                        or (not final and base_p + add_p >= 2**32 - 64) or (final and int(off_v.max()) + int(n_vis.max()) >= 2**32 - 64) \
                        or base_n + add_n >= 2**32 \
                        or int(n_near.max()) > 65535:
                    return None
                cell_rows[rows] = torch.stack([off_v, n_vis, off_n, cnt[:, 1] + (n_near << 16)], 1)
                hdr = _as_u32(cell_rows)
                # (final level: vis entries of 32 bits - wall and view arc; a parent level: bare 16-bit wall numbers)
                pool = torch.zeros(add_p + 64, dtype=torch.int32 if final else torch.int16, device=dev)
                near = torch.zeros((max(add_n, 1), 4), dtype=torch.float32, device=dev) if final else None
                struct.wg_cells = hdr.data_ptr()
                # the fill kernel writes at the headers' offsets: hand it this group's pools displaced by what came before
                _lib.check(h.ms_wallgrid_fill(C.byref(struct), r32.data_ptr(), len(r), mc, bits_starts.data_ptr(), bits.data_ptr(),
                                              None if final else C.c_void_p(pool.data_ptr() - 2*base_p),
                                              C.c_void_p(pool.data_ptr() - 4*base_p) if final else None,      # (+ wg_pool_base[n], in the kernel)
                                              C.c_void_p(near.data_ptr() - 16*base_n) if final else None, _stream(dev)))
                torch.cuda.current_stream(dev).synchronize()           # (hdr / bits / pools of this group are done with)
                pools.append(pool[:add_p])
                if final:
                    nears.append(near[:add_n])
                base_p, base_n = base_p + add_p, base_n + add_n
        pool = torch.cat(pools + [torch.zeros(64, dtype=pools[0].dtype, device=dev)])
        near = torch.cat(nears + [torch.zeros((1, 4), dtype=torch.float32, device=dev)]) if final else None
        pool_base = pool_base[rep].contiguous()                       # (members: their representative's)
        return _as_u32(cell_rows), starts, geom, pool, near, cells.to(torch.int32), pool_base

    def _wall_checksum(self):
        """A 64-bit checksum of the static walls' rows as they are now (their bits, position-weighted, summed modulo 2^64):
        what the wall grid was built from, if taken when it was. (The weights - zero for the agents' rows, which every
        render rewrites - are made once per scenery: the check itself is one multiply-and-sum over the lines.)"""
        ln = self._lines
        if self._wg_weights is None:
            af = self._n_agents*self._model.shape[0]
            k = torch.arange(ln.vals.shape[0], device=ln.vals.device)
            static = (k - ln.starts.long()[ln.inverse.long()]) >= af
            self._wg_weights = ((8*k[:, None] + 2*torch.arange(4, device=k.device)[None] + 1)*0x9E3779B1)*static[:, None]   # odd, distinct per word
        return int((ln.vals.reshape(-1, 4).view(torch.int32)*self._wg_weights).sum())

    def check_wall_grid(self):
        """Raises if the static walls are no longer the ones the wall grid was built from (see :func:`bake`): rays and
        collisions that walk a stale grid silently miss the walls that moved. One reduction over the lines and a sync -
        switched on for every :func:`render` / :func:`physics` call by ``MEGASTEP_CHECK_GRID=1`` (as the test suite runs)."""
        if self._wg is not None and self._wg_sum != self._wall_checksum():
            raise RuntimeError('static walls have been changed since cuda.bake() built the wall grid from them: bake again '
                               '(or bake(wall_grid=False) to go without a grid)')

    def _build_wall_grid(self):
        """Builds the wall grid from the static walls as they are now (called by :func:`bake`): per floorplan a uniform
        grid, per cell the walls a ray from the cell can be decided by and the walls an agent in it can run into - first
        for cells WALL_GRID_COARSE times the size, whose lists are all that the cells proper then look at. Installs ``_wg``."""
        self._wg, self._struct = None, None
        self._wg_sum = None
        self._wg_report = None
        if not self.WALL_GRID:
            return
        ln = self._lines
        if len(ln) == 0 or int((ln.widths.long() - self._n_agents*self._model.shape[0]).max()) <= 0:
            return
        lo, hi = self._wall_bounds()
        walls = (ln.widths.long() - self._n_agents*self._model.shape[0]).clamp(min=0)
        for cell in (self.WALL_GRID_CELL, 2*self.WALL_GRID_CELL, 4*self.WALL_GRID_CELL):
            # which envs get a grid at all: some walls, not too many for 16-bit indices, not too large an area
            extent = torch.ceil((hi - lo + 1.5)/cell).clamp(min=1)
            usable = (extent[:, 0]*extent[:, 1] <= self.WALL_GRID_MAX_CELLS) & (walls > 0) & (walls <= 65535)
            parent = None
            if self.WALL_GRID_COARSE:
                try:
                    level = self._scan_level(cell*self.WALL_GRID_COARSE, None, False, usable)
                except torch.cuda.OutOfMemoryError:
                    level = None
                    torch.cuda.empty_cache()
                if level is not None:
                    hdr, starts, geom, pool, _, cells, _ = level
                    parent = (hdr, starts, geom, float(cell*self.WALL_GRID_COARSE), pool, cells)
            try:
                level = self._scan_level(cell, parent, True, usable)
            except torch.cuda.OutOfMemoryError:
                # (the budget is an estimate of the final size; the build's peak is higher - on a device that is nearly full the
                # next coarser level is the answer, not a crash)
                level, parent = None, None
                torch.cuda.empty_cache()
            if level is not None:
                hdr, starts, geom, pool, near, _, pool_base = level
                self._wg = (hdr, starts, geom, float(cell), float(self.WALL_GRID_REACH[0]), float(self.WALL_GRID_REACH[1]),
                            float(self.WALL_GRID_NEAR), pool, near, pool_base)
                self._wg_sum = self._wall_checksum()
                self._struct = None
                size = sum(t.numel()*t.element_size() for t in (hdr, starts, geom, pool, near, pool_base))
                arange = torch.arange(len(ln), device=hdr.device)
                reps = int((usable & ((self._geom.long() if self._geom is not None else arange) == arange)).sum())
                self._wg_report = dict(bytes=size, bytes_per_floorplan=size/max(reps, 1), cell=float(cell), cells=int(hdr.shape[0] - 1),
                                       envs=int(usable.sum()), floorplans=reps,
                                       vis_entries=int(pool.numel() - 64), near_rows=int(near.shape[0] - 1), budget=self._wall_grid_budget(),
                                       coarsened=cell != self.WALL_GRID_CELL)
                if os.environ.get('MEGASTEP_VERBOSE') or cell != self.WALL_GRID_CELL:
                    # (a grid that had to coarsen costs the step 3-6 %: never silently)
                    print(f'megastep_amd: wall grid of {size/2**20:.0f} MiB for {int(usable.sum())} envs ({reps} floorplans) at {cell:g} m cells'
                          + (f' - {self.WALL_GRID_CELL:g} m cells would not fit WALL_GRID_BYTES = {self._wall_grid_budget()/2**30:.1f} GiB'
                             if cell != self.WALL_GRID_CELL else ''), file=sys.stderr, flush=True)
                return
        self._wg_report = dict(bytes=0, cell=None, budget=self._wall_grid_budget(), note='no env a grid could be built for, or none within the budget')
        print(f'megastep_amd: no wall grid built (no env it could serve, or not even {4*self.WALL_GRID_CELL:g} m cells within WALL_GRID_BYTES = '
              f'{self._wall_grid_budget()/2**30:.1f} GiB): every ray and agent meets every wall of its env', file=sys.stderr, flush=True)

    def _bake_plan(self):
        """Scratch for the two-phase bake (MsScenery.bake_vis): for each representative env, lights x ceil(texels/64)
        words. Returns (vis, starts) - torch tensors that must outlive the launch."""
        li, ln, tx = self._lights, self._lines, self._textures
        dev = ln.vals.device
        n_envs = len(ln)
        first, last = ln.starts.long(), (ln.ends - 1).long().clamp(min=0)
        texels = torch.where(ln.widths > 0, tx.ends.long()[last] - tx.starts.long()[first], torch.zeros_like(first))
        words = li.widths.long()*((texels + 63)//64)
        rep = torch.arange(n_envs, device=dev) if self._geom is None else self._geom.long()
        own = words*(rep == torch.arange(n_envs, device=dev))
        starts = (own.cumsum(0) - own)[rep].contiguous()
        # (zeroed: a row that visibility_kernel's bounds check skipped reads as 'nothing blocks', never as garbage)
        return torch.zeros(max(int(own.sum()), 1), dtype=torch.int64, device=dev), starts

    def _device(self):
        """The GPU all of this scenery's tensors live on (checked once; the tensors cannot be swapped out)."""
        if self._dev is None:
            self._dev = _require_gpu(*self._tensors())
        return self._dev

    def _tensors(self):
        li, ln, tx = self._lights, self._lines, self._textures
        return (ln.vals, ln.widths, li.vals, li.widths, tx.vals, tx.widths, self._model, self._baked.vals)


class Render:
    """Result of :func:`render` (reference: common.h:216-222, wrappers.cpp:147-164). Fields that were not asked for
    are ``None``; ``obs_rgb``/``obs_depth`` are the pooled observations when the call asked for them."""

    def __init__(self, indices, locations, dots, distances, screen, obs_rgb=None, obs_depth=None, obs_subsample=None,
                 obs_centre=None):
        self._t = (indices, locations, dots, distances, screen)
        self._obs = (obs_rgb, obs_depth, obs_subsample, obs_centre)

    indices = property(lambda self: self._t[0])
    locations = property(lambda self: self._t[1])
    dots = property(lambda self: self._t[2])
    distances = property(lambda self: self._t[3])
    screen = property(lambda self: self._t[4])
    obs_rgb = property(lambda self: self._obs[0])
    obs_depth = property(lambda self: self._obs[1])
    obs_subsample = property(lambda self: self._obs[2])
    obs_centre = property(lambda self: self._obs[3])


class Physics:
    """Result of :func:`physics` (reference: common.h:224-226, wrappers.cpp:166-172)."""

    def __init__(self, progress):
        self._progress = progress

    progress = property(lambda self: self._progress)


# ---------------------------------------------------------------------------------------------------------------------
# kernels
# ---------------------------------------------------------------------------------------------------------------------
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream(dev):
    # (the current stream's handle straight from torch's C side: torch.cuda.current_stream() builds a Stream object around it
    # first - 5 us of the host's 25 per launch, three launches a step)
    if _raw_stream is not None and dev.index is not None:
        return C.c_void_p(_raw_stream(dev.index))
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _as_u32(t):
    """An int64 tensor of values below 2^32 as the int32 tensor whose bits are those values as uint32."""
    return torch.where(t >= 2**31, t - 2**32, t).to(torch.int32).contiguous()


def _memory_share(dev, fraction):
    """The default byte budget of a grid: 1/`fraction` of the device's memory - but never more than a third of what is FREE on
    it right now (next to a policy and its optimizer, or on a small GPU, a share of the TOTAL is memory that is not there:
    ADVICE r5; building a wall grid peaks at about twice its final size, so a third of the free memory is what can be afforded)."""
    total = torch.cuda.get_device_properties(dev).total_memory
    try:
        free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)   # (+ what torch's cache holds idle)
    except Exception:
        free = total
    return int(min(total//fraction, free//3))


class _on:
    """Makes ``dev`` the current HIP device for the launch if it is not already."""

    def __init__(self, dev):
        self._guard = None if dev.index == torch.cuda.current_device() else torch.cuda.device(dev)

    def __enter__(self):
        if self._guard is not None:
            self._guard.__enter__()

    def __exit__(self, *exc):
        if self._guard is not None:
            self._guard.__exit__(*exc)


def _agents_on(agents, dev):
    if agents._dev != dev:
        if agents._dev is None or not agents._dev.type == 'cuda':
            raise RuntimeError('megastep_amd kernels need GPU (HIP) tensors; the agents are on ' + str(agents._dev or 'several devices'))
        raise RuntimeError(f'all tensors must live on one device; got {agents._dev} and {dev}')


def bake(scenery, scratch=True, wall_grid=True):
    """Pre-computes the static lighting of every texel into ``scenery.baked`` (reference: wrappers.cpp:61,
    kernels.cu:270-293). ``scratch=False`` selects the library's self-contained one-pass kernel (no temporary
    allocation, no sharing between envs; same result). Baking also (re)builds the scenery's wall grid - the per-cell
    lists of walls that :func:`render` and :func:`physics` walk instead of every line of the env (include/megastep_hip.h,
    ``MsScenery.wg_*``) - from the walls as they are now: like the baked light it goes stale if static walls are moved
    afterwards (bake again, or pass ``wall_grid=False`` to go without). ``Scenery.check_wall_grid()`` tells (it raises if
    the walls are not the ones the grid was built from), and ``MEGASTEP_CHECK_GRID=1`` has every :func:`render` and
    :func:`physics` call ask it first. HIP graphs captured before a re-bake hold pointers into the old grid: capture again."""
    dev = scenery._device()
    # bake uses none of the initialize() constants (kernels.cu:238-293), and scene.scenery() calls it before any Core
    # exists, so the config is optional here
    cfg = C.byref(_config) if _config is not None else None
    struct = scenery._as_struct()
    if scratch:
        vis, starts = scenery._bake_plan()
        struct = _lib.MsScenery.from_buffer_copy(struct)
        struct.bake_vis, struct.bake_vis_starts, struct.bake_vis_words = vis.data_ptr(), starts.data_ptr(), vis.shape[0]
    # What it cost goes into grid_report() - from events on the CURRENT stream, read when the report is asked for: no device-wide
    # synchronize here (round 5 had three: every other stream of the device stalled for a number nobody may want, and a bake
    # inside a stream capture raised).
    with _on(dev):
        timed = not torch.cuda.is_current_stream_capturing()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if timed else None
        if timed:
            marks[0].record()
        _lib.check(_lib.lib().ms_bake(C.byref(struct), cfg, _stream(dev)))
        if timed:
            marks[1].record()
        if wall_grid:
            scenery._build_wall_grid()
        if timed:
            marks[2].record()
    scenery._bake_marks, scenery._bake_s = (marks, wall_grid), None


def _step_options(agents, shape, movement, respawn, lifespans, imu):
    """``physics``' optional arguments (see there) as the C-ABI's MsMovement / MsStepExtras, by reference (or None)."""
    mv = None
    if movement is not None:
        actions, table, keep = movement
        _check(table, 'movement table', torch.float32, 2)
        if actions.dtype != torch.int64 or not actions.is_contiguous() or actions.shape != agents.angles.shape \
                or table.shape[1:] != (3,) or table.shape[0] < 1:
            raise RuntimeError('movement must be ((N, A) contiguous int64 actions, (K, 3) float32 table, keep)')
        _require_gpu(actions, table)
        mv = C.byref(_lib.MsMovement(actions.data_ptr(), table.data_ptr(), table.shape[0], float(keep)))
    ex = None
    if respawn is not None or lifespans is not None or imu is not None:
        x = _lib.MsStepExtras(imu_ang_scale=1., imu_speed_scale=1.)
        used = []
        if respawn is not None:
            mask, choices = respawn['mask'], respawn['choices']
            positions, angles = respawn['positions'], respawn['angles']
            if mask.dtype != torch.bool or mask.shape != shape or choices.dtype != torch.int64 or choices.shape != shape \
                    or positions.dtype != torch.float32 or positions.ndim != 4 or positions.shape[:2] != shape \
                    or positions.shape[3] != 2 or angles.dtype != torch.float32 or angles.shape != positions.shape[:3] \
                    or not all(t.is_contiguous() for t in (mask, choices, positions, angles)):
                raise RuntimeError('respawn needs a (N, A) bool mask, (N, A) int64 choices, (N, A, S, 2) float32 positions '
                                   'and (N, A, S) float32 angles, all contiguous')
            x.respawn_mask, x.respawn_choice = mask.data_ptr(), choices.data_ptr()
            x.spawn_positions, x.spawn_angles, x.n_spawns = positions.data_ptr(), angles.data_ptr(), positions.shape[2]
            x.respawn_after = int(bool(respawn.get('after', False)))
            used += [mask, choices, positions, angles]
        if lifespans is not None:
            if respawn is None:
                raise RuntimeError('lifespans need a respawn mask to report into')
            ages, maxima, fresh = lifespans['lifespans'], lifespans['max_lifespans'], lifespans['fresh']
            if any(t.dtype != torch.int32 or t.shape != shape or not t.is_contiguous() for t in (ages, maxima, fresh)):
                raise RuntimeError('lifespans, max_lifespans and fresh must be contiguous (N, A) int32 tensors')
            x.lifespans, x.max_lifespans, x.fresh_max = ages.data_ptr(), maxima.data_ptr(), fresh.data_ptr()
            used += [ages, maxima, fresh]
        if imu is not None:
            obs, ang_scale, speed_scale = imu
            if obs.dtype != torch.float32 or obs.shape != shape + (3,) or not obs.is_contiguous():
                raise RuntimeError('the imu output must be a contiguous (N, A, 3) float32 tensor')
            x.imu, x.imu_ang_scale, x.imu_speed_scale = obs.data_ptr(), float(ang_scale), float(speed_scale)
            used.append(obs)
        _require_gpu(*used)
        ex = C.byref(x)
    return mv, ex


def physics(scenery, agents, movement=None, out=None, respawn=None, lifespans=None, imu=None, config=None):
    """Advances the agents by one step, stopping them at walls and at each other; updates ``agents`` in place and
    returns :class:`Physics` with the (N, A) ``progress`` (reference: wrappers.cpp:69, kernels.cu:179-230).

    Beyond the reference, the tensor ops its callers run around the step can ride in the same launch
    (include/megastep_hip.h, MsMovement / MsStepExtras):

    * ``movement=(actions, table, keep)``: the movement modules' velocity update first: ``actions`` (N, A) int64 rows of
      ``table`` (K, 3) = agent-frame [dx, dy, d angvelocity]; velocities become ``keep*old + delta`` (``keep = 0`` assigns);
    * ``respawn=dict(mask, choices, positions, angles, after=False)``: agents marked in the (N, A) bool ``mask`` get
      pose ``positions[n, a, choices[n, a]]`` / ``angles[...]`` and zero velocities - before the step (and before the
      movement), or after it with ``after=True``;
    * ``lifespans=dict(lifespans, max_lifespans, fresh)``: (N, A) int32 ages tick first; agents at their maximum are
      added to ``respawn['mask']`` (required with it), start over and take ``fresh`` as their new maximum;
    * ``imu=(out, ang_scale, speed_scale)``: the (N, A, 3) IMU observation of the state the step leaves behind.

    ``out``: the :class:`Physics` of an earlier call, to write ``progress`` into instead of allocating.
    ``config``: the constants of this call (:func:`config`); default: the ones ``agents`` carry (a Core's do), else initialize()'s."""
    dev = scenery._device()
    _agents_on(agents, dev)
    shape = (len(scenery.lines), scenery.n_agents)
    if agents.angles.shape != shape:
        raise RuntimeError('agents do not match the scenery: expected (n_envs, n_agents) = '
                           f'{shape}, got {tuple(agents.angles.shape)}')
    mv, ex = _step_options(agents, shape, movement, respawn, lifespans, imu)
    progress = torch.empty_like(agents.angles) if out is None else out.progress      # `out`: an earlier call's Physics
    _ab_switches()
    _check_grid(scenery, dev)
    with _on(dev):
        _lib.check(_lib.lib().ms_step_physics(C.byref(scenery._as_struct()), C.byref(agents._struct if agents._use_cache else agents._plain),
                                              mv, ex, C.c_void_p(progress.data_ptr()), C.byref(_cfg(agents, config)), _stream(dev)))
    agents._cached = agents._use_cache
    agents._epoch += 1
    return Physics(progress) if out is None else out


def deathmatch_shoot(centre, positions, upper, health, damage, dead, clearance=1., hit_damage=.05, tick_damage=.001,
                     out=None, matchings=False):
    """The Deathmatch env's game logic between one frame and the next as ONE launch (include/megastep_hip.h, MsDeathmatch;
    reference: demo/envs/deathmatch.py:46-88 - ``_reset`` + ``_shoot`` + the ``health`` observation, some twenty tensor ops).

    ``centre`` (N, A, 2) int32: :func:`render`'s ``obs_centre`` of this frame; ``positions`` (N, A, 2); ``upper`` (N, 2): the
    floorplans' extents + clearance; ``health``, ``damage`` (N, A) float32 and ``dead`` (N, A) bool, all updated IN PLACE:
    agents marked in ``dead`` (the mask this step's physics launch respawned by) start from health 1 / damage 0, then
    everyone takes this frame's hits, wounds and strays, and ``dead`` becomes ``health <= 0`` - the next step's mask.
    Returns ``(reset, reward, health_obs[, matchings])``: the incoming ``dead``, the hits dealt, a copy of the new health -
    fresh tensors, or the ones of an earlier call passed as ``out``."""
    n, a = health.shape
    _check(centre, 'centre', torch.int32, 3); _check(positions, 'positions', torch.float32, 3); _check(upper, 'upper', torch.float32, 2)
    _check(health, 'health', torch.float32, 2); _check(damage, 'damage', torch.float32, 2); _check(dead, 'dead', torch.bool, 2)
    if centre.shape != (n, a, 2) or positions.shape != (n, a, 2) or upper.shape != (n, 2) or damage.shape != (n, a) or dead.shape != (n, a):
        raise RuntimeError('deathmatch_shoot: centre (N, A, 2), positions (N, A, 2), upper (N, 2), health / damage / dead (N, A)')
    dev = _require_gpu(centre, positions, upper, health, damage, dead)
    if out is None:
        out = (torch.empty_like(dead), torch.empty_like(health), torch.empty_like(health)) + \
              ((torch.empty((n, a, a), dtype=torch.bool, device=dev),) if matchings else ())
    dm = _lib.MsDeathmatch(centre.data_ptr(), positions.data_ptr(), upper.data_ptr(), float(clearance), float(hit_damage), float(tick_damage),
                           health.data_ptr(), damage.data_ptr(), dead.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                           out[3].data_ptr() if len(out) > 3 else None)
    with _on(dev):
        _lib.check(_lib.lib().ms_deathmatch_shoot(n, a, C.byref(dm), _stream(dev)))
    return out


def explorer_books(tally, before, lengths, epoch, over, slack, pixels, display=False):
    """The Explorer env's bookkeeping between one frame and the next as ONE launch (include/megastep_hip.h, MsExplorer;
    reference: demo/envs/explorer.py:45-90 - the reward, the counters of ``_reset`` and the episode rule of ``step``, a dozen
    tensor ops). ``tally`` is the first-sight count :func:`render` keeps (``seen=``), ``epoch`` its epochs; ``before``,
    ``lengths`` (N,) int32 and ``over`` (N,) bool are the env's own - all updated IN PLACE: ``over`` comes in as the envs this
    step respawned (they get no reward) and leaves as the envs the next step is to respawn, which have already forgotten what
    they saw. Returns ``(reset, reward)`` - the incoming ``over`` and this frame's reward - plus, with ``display``, the
    potential and the lengths as this step leaves them."""
    n = tally.shape[0]
    for name, t in (('tally', tally), ('before', before), ('lengths', lengths), ('epoch', epoch)):
        _check(t, name, torch.int32, 1)
        if t.shape != (n,):
            raise RuntimeError('explorer_books: tally, before, lengths, epoch and over must all be (N,)')
    _check(over, 'over', torch.bool, 1)
    if over.shape != (n,):
        raise RuntimeError('explorer_books: tally, before, lengths, epoch and over must all be (N,)')
    dev = _require_gpu(tally, before, lengths, epoch, over)
    ptrs = (tally.data_ptr(), before.data_ptr(), lengths.data_ptr(), epoch.data_ptr(), over.data_ptr())
    reset = torch.empty_like(over)
    rest = torch.empty((3 if display else 1, n), dtype=torch.float32, device=dev)         # (one allocation: reward | potential | lengths)
    out = (reset, rest[0]) + ((rest[1], rest[2].view(torch.int32)) if display else ())
    ex = _lib.MsExplorer(*ptrs, int(slack), int(pixels),
                         reset.data_ptr(), rest.data_ptr(), rest.data_ptr() + 4*n if display else None, rest.data_ptr() + 8*n if display else None)
    with _on(dev):
        _lib.check(_lib.lib().ms_explorer_books(n, C.byref(ex), _stream(dev)))
    return out


FIELDS = ('indices', 'locations', 'dots', 'distances', 'screen')
#: MEGASTEP_CHECK_GRID=1: every render / physics call first makes sure the static walls are still the ones the wall grid
#: was built from (a reduction over the lines and a host sync per call - for debugging and the test suite, off by default)
CHECK_GRID = os.environ.get('MEGASTEP_CHECK_GRID', '0') not in ('', '0')


def _check_grid(scenery, dev):
    if CHECK_GRID and scenery._wg is not None:
        with _on(dev):                                                    # (the stream that matters is `dev`'s current one)
            capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            scenery.check_wall_grid()


def step_render(scenery, agents, fields=None, pooled=None, out=None, seen=None, config=None, movement=None, respawn=None,
                lifespans=None, imu=None):
    """One step of the hot path - :func:`physics` then :func:`render`, what every ``env.step()`` of the reference runs
    (wrappers.cpp:69 + :82) - as one call, and where the shapes allow it as ONE LAUNCH (include/megastep_hip.h,
    ``ms_step_render``): with one agent per env and at most 64 rays (BASELINE config 2, the Explorer shape) an agent is a single
    wavefront, which runs its env's physics step and renders from the pose it ends on. Any other shape is the two launches,
    as if the two calls had been made. Same bits either way (tests/test_gpu_step_render.py).

    Arguments as :func:`render`'s, and :func:`physics`' ``movement`` / ``respawn`` / ``lifespans`` / ``imu`` (a whole env.step() of a
    single-agent env of up to 64 rays - the reference's tutorial env - is then one launch); ``out``: the ``(Physics, Render)`` of an
    earlier call, to write into. Returns ``(Physics, Render)``."""
    physics_out, render_out = out if out is not None else (None, None)
    progress = torch.empty_like(agents.angles) if physics_out is None else physics_out.progress
    shape = (len(scenery.lines), scenery.n_agents)
    if agents.angles.shape != shape:
        raise RuntimeError('agents do not match the scenery')
    options = _step_options(agents, shape, movement, respawn, lifespans, imu)
    r = render(scenery, agents, fields=fields, pooled=pooled, out=render_out, seen=seen, config=config, _progress=(progress, *options))
    agents._cached = agents._use_cache
    agents._epoch += 1
    return (Physics(progress) if physics_out is None else physics_out), r


def render(scenery, agents, fields=None, pooled=None, telemetry=False, out=None, seen=None, config=None, _progress=None):
    """Casts ``res`` rays per agent and shades them; also rewrites the agents' model lines in ``scenery.lines``
    (reference: wrappers.cpp:82, kernels.cu:452-475). Returns :class:`Render`.

    Extensions over the reference, all off by default: ``fields`` names the per-ray outputs that are wanted
    (the others are neither written nor allocated; a call that wants no colour - no ``screen``, no pooled RGB - runs an
    instantiation of the kernel without the shading pass: ``fields=('distances',)`` is what ``modules.Depth`` reads,
    reference modules.py:170-184), and ``pooled=dict(subsample=s, max_depth=d, rgb=True, depth=True)``
    has the kernel write the mean-pooled observations of ``modules.RGB`` / ``modules.Depth`` itself
    (``Render.obs_rgb`` (n, a, 3, res/s), ``Render.obs_depth`` (n, a, res/s)). ``out`` takes the :class:`Render` of an
    earlier call with the same arguments and writes into its tensors instead of allocating (the reference allocates
    five tensors per call, kernels.cu:461-469; a caller that consumes a frame before asking for the next need not).
    ``pooled['centre']=True`` adds ``Render.obs_centre`` (n, a, 2) int32: the agent each of the two central observation
    pixels shows, or -1 (all that Deathmatch reads from ``indices``). ``seen=(stamp, epoch, count)`` - int32 tensors
    with one entry per texel, per env and per env - has the kernel do Explorer's first-sight bookkeeping: texels under
    this frame's rays get their env's epoch as stamp, those that did not carry it yet are added to ``count``.
    ``telemetry=True`` (tests) takes the self-contained path whose scratch counters end up in ``Render._telemetry``.

    The contract the wall grid adds: hit indices come from the per-cell lists :func:`bake` made of the
    static walls, so walls must not be moved in place afterwards without baking again - ``Scenery.check_wall_grid()`` /
    ``MEGASTEP_CHECK_GRID=1`` detect it (the reference reads ``lines`` afresh every call and has no such rule)."""
    dev = scenery._device()
    _agents_on(agents, dev)
    n, a = agents.angles.shape
    if (n, a) != (len(scenery.lines), scenery.n_agents):
        raise RuntimeError('agents do not match the scenery')
    cfg = _cfg(agents, config)            # (``config=``: see physics)
    seen_ptrs = None
    if seen is not None:
        stamp, epoch, count = seen
        if any(t.dtype != torch.int32 or not t.is_contiguous() for t in seen) or stamp.shape != (scenery.textures.vals.shape[0],) \
                or epoch.shape != (n,) or count.shape != (n,):
            raise RuntimeError('seen must be contiguous int32 tensors (stamp per texel, epoch per env, count per env)')
        _require_gpu(*seen)
        if a > 1 and scenery._as_struct().lg_vals is None:
            raise RuntimeError('first-sight bookkeeping (seen=) rides in the one-kernel renderer, which needs the light grid that this '
                               'multi-agent scenery was built without (Scenery.LIGHT_GRID = False)')
        seen_ptrs = tuple(t.data_ptr() for t in seen)
    # (what an `out` must have been made for: put together only when one is handed over, or asked of a result later)
    spec = (n, a, cfg.res, fields, pooled, dev, seen_ptrs)
    if out is not None:
        made_for = getattr(out, '_key', None)
        if made_for is None:
            made_for = out._key = _render_key(getattr(out, '_spec', None))   # (once per result)
        if made_for != _render_key(spec):
            raise RuntimeError('`out` must come from a render call with the same shapes, fields and pooling')
        result = out
    else:
        result = _render_buffers(scenery, n, a, cfg.res, fields, pooled, dev)
        result._spec = spec
        if seen is not None:
            result._struct.seen_stamp, result._struct.seen_epoch, result._struct.seen_count = seen_ptrs
    _ab_switches()
    _check_grid(scenery, dev)
    with _on(dev):
        use_cache = agents._cached and not telemetry
        if telemetry:
            _lib.lib().ms_debug_pair_telemetry(1)               # the kernels' pair counters too (tools/pair_stats.py)
        try:
            if _progress is not None:                            # step_render: the physics step first, in the same launch where it can be
                progress, mv, ex = _progress
                _lib.check(_lib.lib().ms_move_step_render(C.byref(scenery._as_struct()), C.byref(agents._struct if agents._use_cache else agents._plain),
                                                          mv, ex, C.c_void_p(progress.data_ptr()), C.byref(result._struct), C.byref(cfg), _stream(dev)))
            else:
                _lib.check(_lib.lib().ms_render(C.byref(scenery._as_struct()), C.byref(agents._struct if use_cache else agents._plain),
                                                C.byref(result._struct), C.byref(cfg), _stream(dev)))
        finally:
            if telemetry:
                _lib.lib().ms_debug_pair_telemetry(0)
    return result


_layouts = {}
def _render_key(spec):
    """A render call's shapes, fields, pooling and books as a comparable value (see ``out=``)."""
    if spec is None:
        return None
    n, a, res, fields, pooled, dev, seen_ptrs = spec
    return (n, a, res, None if fields is None else tuple(fields), None if pooled is None else tuple(sorted(pooled.items())), dev, seen_ptrs)


def _render_buffers(scenery, n, a, r, fields, pooled, dev):
    """One allocation for the wanted outputs (reference: five at::empty calls, kernels.cu:461-469), the pooled
    observations and the kernels' scratch (MS_RENDER_WORKSPACE_INTS), and the MsRender that points into it."""
    scenery._as_struct()
    lit = a == 1 or scenery._lg[0] is not None
    key = (n, a, r, None if fields is None else tuple(fields), None if pooled is None else tuple(sorted(pooled.items())), lit)
    layout = _layouts.get(key)
    if layout is None:
        layout = _layouts[key] = _render_layout(n, a, r, fields, pooled, lit)
        if len(_layouts) > 64:
            _layouts.pop(next(iter(_layouts)))
    total, offs, sizes, shapes, sub, max_depth, has_pool = layout
    buf = torch.empty(total, dtype=torch.float32, device=dev)
    base = buf.data_ptr()
    pieces = [buf[offs[i]:offs[i] + shapes[i][1]].view(shapes[i][0]) if sizes[i] else None for i in range(8)]
    if pieces[0] is not None:
        pieces[0] = pieces[0].view(torch.int32)
    if pieces[7] is not None:
        pieces[7] = pieces[7].view(torch.int32)
    result = Render(*pieces[:5], pieces[5], pieces[6], sub if has_pool else None, pieces[7])
    ptrs = [base + 4*offs[i] if sizes[i] else None for i in range(8)]
    result._struct = _lib.MsRender(*ptrs[:5], base + 4*offs[-1], ptrs[5], ptrs[6], sub, max_depth, ptrs[7])
    result._telemetry = buf[offs[-1]:offs[-1] + 16].view(torch.int32)     # see render_prep_kernel (which zeroes it); read by the tests
    return result


def _render_layout(n, a, r, fields, pooled, lit):
    """Where everything sits in a render call's one allocation - worked out once per (shapes, fields, pooling) and kept: it is
    a fifth of the host's share of a render call."""
    want = FIELDS if fields is None else tuple(fields)
    if any(f not in FIELDS for f in want):
        raise RuntimeError(f'fields must be among {FIELDS}')
    colour = 'screen' in want or (pooled is not None and pooled.get('rgb', True))
    if not lit and colour:
        # no light grid (Scenery.LIGHT_GRID switched off): agent hits are lit by a second launch that reads all five
        # planes back and patches `screen` - after any pooling. All planes then, and the caller pools.  (Without colour
        # there is nothing to light: the depth-only kernel serves such sceneries like any other.)
        want, pooled = FIELDS, None
    sub, max_depth, w = 1, 1., r
    n_rgb = n_depth = n_centre = 0
    if pooled is not None:
        sub, max_depth = int(pooled.get('subsample', 1)), float(pooled.get('max_depth', 10.))
        if sub < 1 or sub & (sub - 1) or 64 % sub or r % sub:
            raise RuntimeError('pooled subsample must be a power of two dividing 64 and the resolution')
        w = r//sub
        n_rgb = 3*n*a*w if pooled.get('rgb', True) else 0
        n_depth = n*a*w if pooled.get('depth', True) else 0
        n_centre = 2*n*a if pooled.get('centre', False) else 0
    plane = n*a*r
    sizes = [plane*(3 if f == 'screen' else 1) if f in want else 0 for f in FIELDS] + [n_rgb, n_depth, n_centre]
    sizes = [(x + 3) & ~3 for x in sizes]                                # keep every piece 16-byte aligned
    offs = [0]
    for x in sizes:
        offs.append(offs[-1] + x)
    total = offs[-1] + 18 + n*a*((r + 63)//64) + 2*n*a
    shapes = [((n, a, r, 3) if f == 'screen' else (n, a, r)) for f in FIELDS] + [(n, a, 3, w), (n, a, w), (n, a, 2)]
    shapes = [(sh, int(torch.Size(sh).numel())) for sh in shapes]
    return total, offs, sizes, shapes, sub, max_depth, pooled is not None
