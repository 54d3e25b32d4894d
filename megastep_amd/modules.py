"""Chunks of functionality that turn up in most environments (reference: megastep/modules.py:10-381).

These are the direct callers of the hot path: the movement modules end in :func:`cuda.physics`, :func:`render` wraps
:func:`cuda.render`. Everything here is thin torch glue over the tensors the kernels own."""
import numpy as np
import torch
from . import spaces, geometry, cuda, arrdict


def _sincos_deg(angles):
    a = np.pi/180*angles
    return torch.sin(a), torch.cos(a)


def to_local_frame(angles, p):
    """Global-frame vectors -> the agents' local frames."""
    s, c = _sincos_deg(angles)
    x, y = p[..., 0], p[..., 1]
    return torch.stack([c*x + s*y, -s*x + c*y], -1)


def to_global_frame(angles, p):
    """Agent-local vectors -> the global frame."""
    s, c = _sincos_deg(angles)
    x, y = p[..., 0], p[..., 1]
    return torch.stack([c*x - s*y, s*x + c*y], -1)


def _actionset(core, linear, angular):
    # noop, forward/backward, strafe left/right, turn left/right
    velocity = torch.tensor([[0., 0.], [0., 1.], [0., -1.], [1., 0.], [-1., 0.], [0., 0.], [0., 0.]])
    angvelocity = torch.tensor([0., 0., 0., 0., 0., +1., -1.])
    return arrdict.arrdict(velocity=linear/core.fps*velocity, angvelocity=angular/core.fps*angvelocity).to(core.device)


def _table(actionset):
    """The action table as cuda.physics' ``movement`` wants it: one (dx, dy, dangle) row per action."""
    return torch.cat([actionset.velocity, actionset.angvelocity[:, None]], 1).contiguous()


def _move(core, actionset, actions, keep, respawn=None, imu=None, table=None):
    """The velocity update of both movement modules, then physics. On the GPU it is part of the physics launch
    (cuda.physics' ``movement``) - as are, when the env hands them over, the respawn of the agents it wants respawned
    (``respawn``: :meth:`RandomSpawns.draw`) and the IMU observation of the new state (``imu``: the :class:`IMU`
    module, which then returns it from its next call). The tensor ops below are the same arithmetic, and what the
    reference runs."""
    agents = core.agents
    if agents.angles.is_cuda:
        table = _table(actionset) if table is None else table
        reading = None if imu is None else (torch.empty(agents.angles.shape + (3,), device=core.device), imu.ang_scale, imu.speed_scale)
        result = cuda.physics(core.scenery, agents, movement=(actions.long().contiguous(), table, keep), respawn=respawn, imu=reading)
        if imu is not None:
            imu._pending = (reading[0], agents._epoch)          # valid until somebody else touches the agents
        return result
    if respawn is not None and not respawn['after']:
        _respawn(agents, respawn)
    delta = actionset[actions.long()]
    if keep == 0:
        agents.angvelocity[:] = delta.angvelocity
        agents.velocity[:] = to_global_frame(agents.angles, delta.velocity)
    else:
        agents.angvelocity[:] = keep*agents.angvelocity + delta.angvelocity
        agents.velocity[:] = keep*agents.velocity + to_global_frame(agents.angles, delta.velocity)
    result = cuda.physics(core.scenery, agents)
    if respawn is not None and respawn['after']:
        _respawn(agents, respawn)
    return result


def _respawn(agents, request):
    """Applies a :meth:`RandomSpawns.draw` request with tensor ops (what the kernel does inside the physics launch)."""
    reset, choices = request['mask'], request['choices']
    angles = request['angles'].gather(2, choices[..., None]).squeeze(2)
    positions = request['positions'].gather(2, choices[..., None, None].expand(-1, -1, 1, 2)).squeeze(2)
    agents.angles[:] = torch.where(reset, angles, agents.angles)
    agents.positions[:] = torch.where(reset[..., None], positions, agents.positions)
    agents.velocity[:] = torch.where(reset[..., None], torch.zeros_like(agents.velocity), agents.velocity)
    agents.angvelocity[:] = torch.where(reset, torch.zeros_like(agents.angvelocity), agents.angvelocity)
    agents._epoch += 1                  # (a reading an IMU module holds of the state before this is stale now)


class SimpleMovement:

    def __init__(self, core, speed=10, ang_speed=180, n_agents=None):
        """Movement without momentum: seven actions - nothing, forward/backward, strafe left/right, turn left/right
        (reference: modules.py:24-66)."""
        self.core = core
        self._actionset = _actionset(core, speed, ang_speed)
        self._table = _table(self._actionset)
        self.keep = 0.                           # (of the old velocity: none - see cuda.physics' ``movement``)
        self.space = spaces.MultiDiscrete(n_agents or core.n_agents, 7)

    def __call__(self, decision, respawn=None, imu=None):
        """Sets the agents' velocities from ``decision.actions`` ((n_env, n_agent) ints in 0..6), then steps physics.
        ``respawn`` / ``imu``: see :func:`_move`."""
        return _move(self.core, self._actionset, decision.actions, 0., respawn, imu, self._table)


class MomentumMovement:

    def __init__(self, core, accel=5, ang_accel=180, decay=.125, n_agents=None):
        """Movement with momentum: the seven actions accelerate rather than move, and velocity decays by ``decay``
        each step (reference: modules.py:68-118)."""
        self.core = core
        self._actionset = _actionset(core, accel, ang_accel)
        self._table = _table(self._actionset)
        self.decay = decay
        self.keep = 1 - decay
        self.space = spaces.MultiDiscrete(n_agents or core.n_agents, 7)

    def __call__(self, decision, respawn=None, imu=None):
        return _move(self.core, self._actionset, decision.actions, 1 - self.decay, respawn, imu, self._table)


def unpack(d):
    """``cuda`` result objects -> arrdicts with the same attributes (reference: modules.py:120-124)."""
    if isinstance(d, torch.Tensor):
        return d
    return arrdict.arrdict({k: unpack(getattr(d, k)) for k in dir(d) if not k.startswith('_')})


def render(core, observers=None, fields=None, centre=False, seen=None):
    """Calls :func:`cuda.render` and reshapes for torch convs: every field gets a height-1 axis, ``screen`` becomes
    (n_env, n_agent, 3, 1, res) (reference: modules.py:126-136).

    Beyond the reference: pass the :class:`RGB` / :class:`Depth` modules that will consume the result as ``observers``
    and their mean-pooled observations come straight out of the render kernel (they pick them up from the result
    instead of running a chain of tensor ops over the full-resolution outputs), and name in ``fields`` the
    full-resolution outputs that are still needed (default: all five) - the others are not even written.
    ``centre=True`` adds ``centre`` (n_env, n_agent, 2): the agent in each of the two central observation pixels, or -1
    (needs observers); ``seen`` is passed on to :func:`cuda.render` (first-sight texel bookkeeping)."""
    pooled = None
    if observers:
        pooled = _pooling(tuple(observers), bool(centre))
    raw = cuda.render(core.scenery, core.agents, fields=fields, pooled=pooled, seen=seen)
    return _frame(raw, pooled)


def move_render(core, mover, decision, observers=None, fields=None, centre=False, seen=None, respawn=None, imu=None):
    """A movement module's step and :func:`render` as ONE call - ``mover(decision, respawn=, imu=)`` then ``render(core, ...)``, same
    arguments, same result - through :func:`cuda.step_render`: for a single-agent world of up to 64 rays (the reference's tutorial
    env, demo/envs/minimal.py) a whole ``env.step()`` is then one launch; any other shape is the two launches it always was."""
    agents = core.agents
    pooled = _pooling(tuple(observers), bool(centre)) if observers else None
    reading = None if imu is None else (torch.empty(agents.angles.shape + (3,), device=core.device), imu.ang_scale, imu.speed_scale)
    _, raw = cuda.step_render(core.scenery, agents, fields=fields, pooled=pooled, seen=seen,
                              movement=(decision.actions.long().contiguous(), mover._table, mover.keep), respawn=respawn, imu=reading)
    if imu is not None:
        imu._pending = (reading[0], agents._epoch)
    return _frame(raw, pooled)


def _frame(raw, pooled):
    """A :class:`cuda.Render` as the arrdict the observation modules read (see :func:`render`)."""
    r = arrdict.arrdict({k: getattr(raw, k).unsqueeze(2) for k in cuda.FIELDS if getattr(raw, k) is not None})
    if 'screen' in r:
        r['screen'] = r.screen.permute(0, 1, 4, 2, 3)
    if pooled is not None:
        r['pooled_subsample'] = pooled['subsample']
        if raw.obs_rgb is not None:
            r['pooled_rgb'] = raw.obs_rgb.unsqueeze(3)                       # (n_env, n_agent, 3, 1, res/subsample)
        if raw.obs_depth is not None:
            r['pooled_depth'] = raw.obs_depth.unsqueeze(2).unsqueeze(3)      # (n_env, n_agent, 1, 1, res/subsample)
            r['pooled_max_depth'] = pooled['max_depth']
        if raw.obs_centre is not None:
            r['centre'] = raw.obs_centre
    return r


_poolings = {}


def _pooling(observers, centre):
    """What :func:`cuda.render` is asked to pool for these observers - worked out once per set of them (an env hands over the
    same modules every step)."""
    key = (tuple((id(o), o.subsample, getattr(o, 'max_depth', None)) for o in observers), centre)
    pooled = _poolings.get(key)
    if pooled is None:
        subs = {o.subsample for o in observers}
        depth = [o for o in observers if isinstance(o, Depth)]
        if len(subs) != 1 or len({o.max_depth for o in depth}) > 1:
            raise ValueError('observers of one render must share their subsample and max_depth')
        pooled = dict(subsample=subs.pop(), max_depth=depth[0].max_depth if depth else 10.,
                      rgb=any(isinstance(o, RGB) for o in observers), depth=bool(depth), centre=centre)
        if len(_poolings) > 256:
            _poolings.clear()
        _poolings[key] = pooled
    return pooled


def downsample(screen, subsample):
    """(..., W) -> (..., W/subsample, subsample); chase it with a mean/min/max over the last axis
    (reference: modules.py:138-145)."""
    return screen.view(*screen.shape[:-1], screen.shape[-1]//subsample, subsample)


class Depth:

    def __init__(self, core, n_agents=None, subsample=1, max_depth=10):
        """Depth observations in [0, 1]: one at the near plane, zero at ``max_depth`` metres
        (reference: modules.py:147-189)."""
        self.core = core
        self.space = spaces.MultiImage(n_agents or core.n_agents, 1, 1, core.res//subsample)
        self.max_depth = max_depth
        self.subsample = subsample

    def __call__(self, r=None):
        r = render(self.core) if r is None else r
        if 'pooled_depth' in r and r.pooled_subsample == self.subsample and r.pooled_max_depth == self.max_depth:
            self._last_obs = r.pooled_depth                  # the render kernel has done it (see render())
            return self._last_obs
        depth = 1 - ((r.distances - self.core.agent_radius)/self.max_depth).clamp(0, 1)
        self._last_obs = downsample(depth, self.subsample).mean(-1).unsqueeze(3)
        return self._last_obs

    def state(self, e=0):
        return self._last_obs[e].clone()


class RGB:

    def __init__(self, core, n_agents=None, subsample=1):
        """Linear-RGB observations, (n_env, n_agent, 3, 1, res/subsample) (reference: modules.py:191-238)."""
        self.core = core
        self.space = spaces.MultiImage(n_agents or core.n_agents, 3, 1, core.res//subsample)
        self.subsample = subsample

    def __call__(self, r=None):
        r = render(self.core) if r is None else r
        if 'pooled_rgb' in r and r.pooled_subsample == self.subsample:
            self._last_obs = r.pooled_rgb                    # the render kernel has done it (see render())
            return self._last_obs
        self._last_obs = downsample(r.screen, self.subsample).mean(-1)
        return self._last_obs

    def state(self, e=0):
        return self._last_obs[e].clone()


class IMU:

    def __init__(self, core, speed_scale=10., ang_scale=360., n_agents=None):
        """(angular, medial, lateral) velocity observations, (n_env, n_agent, 3) (reference: modules.py:240-270)."""
        self.core = core
        self.space = spaces.MultiVector(n_agents or core.n_agents, 3)
        self.speed_scale = speed_scale
        self.ang_scale = ang_scale
        # a reading the physics launch has already taken (see _move), with the agents' epoch at that moment: every
        # physics call and every respawn through this module's helpers moves the epoch on, and a reading from an earlier
        # epoch is dropped (state changed behind the modules' back - writes straight into the tensors - is not seen)
        self._pending = None

    def __call__(self):
        agents = self.core.agents
        if self._pending is not None:
            (reading, epoch), self._pending = self._pending, None
            if epoch == agents._epoch:
                return reading
        return torch.cat([
            agents.angvelocity[..., None]/self.ang_scale,
            to_local_frame(agents.angles, agents.velocity)/self.speed_scale], -1)


def random_empty_positions(geometries, n_agents, n_points):
    """(n_geometries, n_agents, n_points, 2) randomly chosen free-cell centres, precomputed so respawns are cheap
    (reference: modules.py:272-296; consumes the global ``np.random`` in the same order, env by env). The free cells of
    a geometry that turns up many times are looked up once."""
    free_cells = {}
    points = np.empty((len(geometries), n_agents, n_points, 2))
    for e, g in enumerate(geometries):
        free = free_cells.get(id(g))
        if free is None:
            free = free_cells[id(g)] = np.argwhere(g['masks'] > 0)           # (row, col) of every free cell
        # two draws from the global stream per env, as the reference makes them: which cells (one row of the table per
        # agent-tuple, at most as many rows as the plan has room for), then the order the table is served in
        rows = max(min(len(free)//n_agents, n_points), 0)
        picks = np.random.choice(np.arange(len(free)), (rows, n_agents), replace=True)
        # a plan too small for n_points distinct rows repeats its table; the last n_points rows are the ones kept
        repeats = int(n_points/rows + 1)
        table = np.tile(free[picks], (repeats, 1, 1))[-n_points:]
        table = table[np.random.permutation(len(table))]
        points[e] = np.swapaxes(geometry.centers(table, g['masks'].shape, g['res']), 0, 1)
    return points


def _device_empty_positions(geometries, n_agents, n_points, device):
    """:func:`random_empty_positions` drawn on the device: every spawn point an independent uniform pick among its
    geometry's free cells (what the reference's choice-then-permute amounts to whenever a geometry has at least
    ``n_agents*n_points`` free cells), from torch's generator instead of ``np.random``."""
    tables, which = {}, np.empty(len(geometries), np.int64)
    for e, g in enumerate(geometries):
        if id(g) not in tables:
            free = np.stack((g['masks'] > 0).nonzero(), -1)
            tables[id(g)] = (len(tables), geometry.centers(free, g['masks'].shape, g['res']))
        which[e] = tables[id(g)][0]
    centres = [c for _, c in tables.values()]
    counts = torch.as_tensor([len(c) for c in centres], device=device)
    starts = (counts.cumsum(0) - counts)[torch.as_tensor(which, device=device)]
    counts = counts[torch.as_tensor(which, device=device)]
    pick = (torch.rand((len(geometries), n_agents, n_points), device=device, dtype=torch.float64)*counts[:, None, None]).long()
    pick = torch.minimum(pick, counts[:, None, None] - 1) + starts[:, None, None]
    return torch.as_tensor(np.concatenate(centres), device=device).float()[pick]


class RandomSpawns:

    def __init__(self, geometries, core, n_spawns=100, fast=False):
        """Respawns agents at random free points of their geometry (reference: modules.py:298-326). ``fast=True``
        draws the spawn tables on the device (same distribution, torch's random stream instead of numpy's) - for
        worlds of 10^4 envs and up, where the reference's per-env loop takes seconds."""
        self.core = core
        if fast:
            positions = _device_empty_positions(geometries, core.n_agents, n_spawns, core.device)
            angles = torch.empty(positions.shape[:3], device=core.device).uniform_(-180, 180)
            self._spawns = arrdict.arrdict(positions=positions, angles=angles)
            return
        positions = random_empty_positions(geometries, core.n_agents, n_spawns)
        angles = core.random.uniform(-180, +180, (len(geometries), core.n_agents, n_spawns))
        self._spawns = arrdict.torchify(arrdict.arrdict(positions=positions, angles=angles)).to(core.device)

    def draw(self, reset, after=False):
        """The respawn of the agents marked in the (n_env, n_agent) bool mask ``reset`` as a request that
        :func:`cuda.physics` (through the movement modules' ``respawn=``) carries out inside its launch - before the
        step, or after it with ``after=True``. Same draw as :meth:`__call__`."""
        return dict(mask=reset.contiguous(), choices=self._choices(reset), positions=self._spawns.positions, angles=self._spawns.angles, after=after)

    #: steps' worth of spawn choices drawn per torch.randint call (see _choices)
    DRAW_AHEAD = 64

    def _choices(self, reset):
        """This step's spawn choice for every agent: uniform among the first ``spawns.shape[1]`` spawn points, from torch's
        generator - drawn DRAW_AHEAD steps at a time and handed out a step's slice per call: on (4096, 1) tensors the draw is a
        launch of its own that lasts as long as the env's whole bookkeeping kernel, every step, for numbers that a handful of
        agents in thousands ever look at. Inside a stream capture the draw stays in the step (a captured slice would be the
        same numbers at every replay; torch.randint is graph-safe)."""
        n = self._spawns.angles.shape[1]
        if reset.is_cuda and not torch.cuda.is_current_stream_capturing():
            ahead = getattr(self, '_ahead', None)
            if ahead is None or self._ahead_at >= len(ahead) or ahead.shape[1:] != reset.shape:
                ahead = self._ahead = torch.randint(0, n, (self.DRAW_AHEAD,) + tuple(reset.shape), device=reset.device)
                self._ahead_at = 0
            self._ahead_at += 1
            return ahead[self._ahead_at - 1]
        return torch.randint(0, n, reset.shape, device=reset.device)

    def __call__(self, reset):
        """``reset`` is an (n_env, n_agent) bool mask; the marked agents get a new pose and zero velocity.

        Same draw as the reference (a uniform choice among the first ``spawns.shape[1]`` spawn points), but made for
        every agent and applied through the mask, so the step needs no ``nonzero`` and with it no host sync."""
        _respawn(self.core.agents, self.draw(reset))


class RandomLifespans:

    def __init__(self, core, max_lifespan, min_lifespan=None):
        """Flags agents that outlive a randomly drawn lifespan, so synchronous envs drift apart
        (reference: modules.py:328-381)."""
        self.min_lifespan = max_lifespan//2 if min_lifespan is None else min_lifespan
        self.max_lifespan = max_lifespan
        self._max_lifespans = torch.zeros((core.n_envs, core.n_agents), dtype=torch.int, device=core.device)
        self._lifespans = torch.zeros_like(self._max_lifespans)
        self._reset(core.agent_full(True))

    def _reset(self, reset):
        self._lifespans.masked_fill_(reset, 0)
        fresh = torch.randint_like(self._max_lifespans, self.min_lifespan, self.max_lifespan)
        self._max_lifespans[:] = torch.where(reset, fresh, self._max_lifespans)

    def __call__(self, reset=None):
        self._lifespans += 1
        reset = torch.zeros_like(self._lifespans, dtype=torch.bool) if reset is None else reset
        reset = (self._lifespans >= self._max_lifespans) | reset
        self._reset(reset)
        return reset

    def state(self, e):
        return arrdict.arrdict(lifespan=self._lifespans[e], max_lifespans=self._max_lifespans[e]).clone()
