"""Exploration: one agent per floorplan, rewarded for every texel of wall it lays eyes on for the first time since its
last respawn. An episode ends when its length reaches 200 steps plus the number of texels seen, so good explorers
live longer. Behaviour follows megastep/demo/envs/explorer.py:8-115.

The reference re-counts every texel of every env each step (a ``scatter_add`` over all of them, explorer.py:45-58).
:class:`SeenTexels` keeps the same tally incrementally and without ever syncing with the host, which is what lets a
whole ``step`` be captured in a HIP graph."""
import torch

from ... import arrdict, core, cubicasa, dotdict, modules, scene

EPISODE_SLACK = 200     # steps an agent gets on top of one per texel seen


def texels_hit(scenery, frame):
    """(n_envs, n_agents, 1, res) texel under each ray of a render result with ``indices`` and ``locations``; -1 for the
    rays that missed (explorer.py:34-43). The render kernel works the same index out for itself when it keeps the
    :class:`SeenTexels` books; this is the tensor-op statement of it."""
    hit = frame.indices >= 0
    line = (scenery.lines.starts[:, None, None, None] + frame.indices.clamp(min=0)).long()
    width = scenery.textures.widths[line].float()
    along = torch.min(torch.floor(width*frame.locations), width - 1)                 # explorer.py:38-41
    texel = scenery.textures.starts[line].long() + torch.where(hit, along, torch.zeros_like(along)).long()
    return torch.where(hit, texel, torch.full_like(texel, -1))


def _plan_workers():
    """Processes that generate the floorplans nobody handed over (numpy-only subprocesses: safe beside an initialised GPU)."""
    import os
    return min(os.cpu_count() or 1, 32)


class SeenTexels:
    """Which texels each env has seen since its last respawn, and how many.

    A texel counts as seen while its stamp equals its env's epoch; a respawn bumps the env's epoch, which forgets all
    of its texels at once without touching them. The render kernel keeps the books as it shades (``cuda.render``'s
    ``seen``): it stamps the texel under every ray and adds the ones that were not stamped yet to ``tally``."""

    def __init__(self, scenery, n_envs):
        device = scenery.textures.vals.device
        self.texel_env = scenery.lines.inverse[scenery.textures.inverse.long()].long()       # texel -> env
        self.epoch = torch.ones(n_envs, dtype=torch.int32, device=device)
        self.stamp = torch.zeros(len(self.texel_env), dtype=torch.int32, device=device)
        # per env: texels seen since the last respawn, the same at the last call of gained(), and a spare row the env
        # keeps its episode length in - one tensor, so that a respawn clears all three with one masked fill
        self.counters = torch.zeros((3, n_envs), dtype=torch.int32, device=device)
        self.tally, self._before, self.spare = self.counters

    #: what a render call needs to keep the books
    books = property(lambda self: (self.stamp, self.epoch, self.tally))
    #: texels seen per env since its last respawn, as the reference's float potential
    count = property(lambda self: self.tally.float())

    def gained(self):
        """How many texels each env saw for the first time since the last call (or its last respawn)."""
        new = self.tally - self._before
        self._before.copy_(self.tally)
        return new

    def forget(self, envs):
        """Envs marked in the bool mask start over."""
        self.epoch += envs
        self.counters.masked_fill_(envs, 0)

    def mask(self):
        """Per texel: has its env seen it since the env's last respawn."""
        return self.stamp == self.epoch[self.texel_env]


class Explorer:

    def __init__(self, n_envs, *args, device='cuda', geometries=None, depth_only=False, fused=None, **kwargs):
        """``depth_only=True``: observations are depth + IMU, no RGB (BASELINE config 2's "64-ray depth-only"); the
        renderer then runs without its shading pass. ``fused`` (default: on a GPU): the bookkeeping between frames - the
        reward, the episode rule, the forgetting - is ONE launch behind the render kernel
        (:func:`megastep_amd.cuda.explorer_books`) instead of a dozen tensor ops; ``False`` keeps the tensor ops, the same
        arithmetic."""
        if geometries is None:
            geometries = cubicasa.sample(n_envs, workers=_plan_workers(), context='subprocess')       # (4096 plans: 3 s on worker processes, 30 in this one)
        self.core = core.Core(scene.scenery(geometries, 1, device=device), *args, res=4*64, fov=130, **kwargs)
        c = self.core
        self.device = c.device

        self._mover = modules.MomentumMovement(c)
        self._respawner = modules.RandomSpawns(geometries, c)
        self._rgb = None if depth_only else modules.RGB(c, n_agents=1, subsample=4)
        self._depth = modules.Depth(c, n_agents=1, subsample=4)
        self._imu = modules.IMU(c)
        self.action_space = self._mover.space
        self.obs_space = dotdict.dotdict(d=self._depth.space, imu=self._imu.space)
        if not depth_only:
            self.obs_space['rgb'] = self._rgb.space

        self._memory = SeenTexels(c.scenery, c.n_envs)
        self._lengths = self._memory.spare                                  # (cleared together with the books)
        self._fused = c.device.type == 'cuda' if fused is None else bool(fused)
        # (fused) who the next step is to respawn - written by the kernel at the end of a step, read by the next physics launch
        # as its respawn mask; the potential and the lengths as the last step left them (the kernel has by then moved the
        # counters themselves on to the next step's top)
        self._over = c.env_full(False)
        self._shown = None

    # what the tests and `state` look at
    _potential = property(lambda self: self._memory.count if self._shown is None else self._shown[0])
    # (fused: an env marked over has already moved its epoch on - until its respawn it is shown what it saw under the old one)
    _seen = property(lambda self: self._memory.mask() if self._shown is None else
                     self._memory.stamp == (self._memory.epoch - self._over.int())[self._memory.texel_env])
    _tex_to_env = property(lambda self: self._memory.texel_env)

    def _restart(self, which):
        self._respawner(which.unsqueeze(-1))
        self._memory.forget(which)

    def _world(self, reset):
        # pooled RGB-D straight from the render kernel, which also keeps the books of the texels it sees: no per-ray
        # output is needed at all
        observers = (self._depth,) if self._rgb is None else (self._rgb, self._depth)
        frame = modules.render(self.core, observers=observers, fields=(), seen=self._memory.books)
        pixels = self.core.res//self._depth.subsample
        reward = (self._memory.gained()/pixels).masked_fill_(reset, 0.)        # nothing for the frame after a respawn
        obs = arrdict.arrdict(d=self._depth(frame), imu=self._imu())
        if self._rgb is not None:
            obs['rgb'] = self._rgb(frame)
        return arrdict.arrdict(obs=obs, reset=reset, reward=reward)

    def _world_fused(self):
        """Render (which keeps the first-sight books), then the reward, the episode rule and the forgetting as one launch."""
        from ... import cuda
        observers = (self._depth,) if self._rgb is None else (self._rgb, self._depth)
        frame = modules.render(self.core, observers=observers, fields=(), seen=self._memory.books)
        m = self._memory
        reset, reward, potential, lengths = cuda.explorer_books(m.tally, m._before, self._lengths, m.epoch, self._over, EPISODE_SLACK,
                                                                self.core.res//self._depth.subsample, display=True)
        self._shown = (potential, lengths)
        obs = arrdict.arrdict(d=self._depth(frame), imu=self._imu())
        if self._rgb is not None:
            obs['rgb'] = self._rgb(frame)
        return arrdict.arrdict(obs=obs, reset=reset, reward=reward)

    @torch.no_grad()
    def reset(self):
        everyone = self.core.env_full(True)
        self._restart(everyone)
        if self._fused:
            self._over.fill_(True)
            return self._world_fused()
        return self._world(everyone)

    @torch.no_grad()
    def step(self, decision):
        if self._fused:
            # four launches: the spawn draw, physics (movement + step + respawn of the envs marked over + IMU), render (pooled
            # observations + first-sight books), the env's books (reward; the next step's episode rule and forgetting)
            self._mover(decision, respawn=self._respawner.draw(self._over.unsqueeze(-1), after=True), imu=self._imu)
            return self._world_fused()
        # Who is over does not depend on this step's movement (explorer.py:83-90 moves first, then checks), so it is
        # settled up front and the respawn rides in the physics launch, after the integration - as does the IMU reading.
        self._lengths += 1
        over = self._lengths >= self._memory.tally + EPISODE_SLACK
        self._mover(decision, respawn=self._respawner.draw(over.unsqueeze(-1), after=True), imu=self._imu)
        self._memory.forget(over)
        return self._world(over)

    def state(self, e=0):
        seen = self._memory.mask()[self._memory.texel_env == e]
        return arrdict.arrdict(core=self.core.state(e), **({} if self._rgb is None else dict(rgb=self._rgb.state(e))), d=self._depth.state(e),
                               potential=self._potential[e].clone(), seen=seen.clone(),
                               length=(self._lengths if self._shown is None else self._shown[1])[e].clone(),
                               max_length=self._potential[e].add(EPISODE_SLACK).clone())
