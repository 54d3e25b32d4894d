"""Exploration: one agent per floorplan, rewarded for every texel of wall it lays eyes on for the first time since its
last respawn. An episode ends when its length reaches 200 steps plus the number of texels seen, so good explorers
live longer. Behaviour follows megastep/demo/envs/explorer.py:8-115.

The reference re-counts every texel of every env each step (a ``scatter_add`` over all of them, explorer.py:45-58).
:class:`SeenTexels` keeps the same tally incrementally and without ever syncing with the host, which is what lets a
whole ``step`` be captured in a HIP graph."""
import torch

from ... import arrdict, core, cubicasa, dotdict, modules, scene

EPISODE_SLACK = 200     # steps an agent gets on top of one per texel seen


class SeenTexels:
    """Which texels each env has seen since its last respawn, and how many.

    A texel counts as seen while its stamp equals its env's epoch; a respawn bumps the env's epoch, which forgets all
    of its texels at once without touching them. ``claim`` settles which of several rays on one texel counts it. One
    extra slot at the end of both arrays takes the rays that hit nothing."""

    def __init__(self, scenery, n_envs):
        device = scenery.textures.vals.device
        self.texel_env = scenery.lines.inverse[scenery.textures.inverse.long()].long()       # texel -> env
        n_texels = len(self.texel_env)
        self.nowhere = torch.tensor(n_texels, device=device)
        self.epoch = torch.ones(n_envs, dtype=torch.int32, device=device)
        self.stamp = torch.zeros(n_texels + 1, dtype=torch.int32, device=device)
        self.claim = torch.full((n_texels + 1,), -1, dtype=torch.long, device=device)
        self.count = torch.zeros(n_envs, device=device)
        self._scenery = scenery

    def texels_hit(self, frame):
        """(n_envs, n_agents, 1, res) texel under each ray of a render result; ``nowhere`` for the rays that missed."""
        sc = self._scenery
        hit = frame.indices >= 0
        line = (sc.lines.starts[:, None, None, None] + frame.indices.clamp(min=0)).long()
        width = sc.textures.widths[line].float()
        along = torch.min(torch.floor(width*frame.locations), width - 1)                 # explorer.py:38-41
        texel = sc.textures.starts[line].long() + torch.where(hit, along, torch.zeros_like(along)).long()
        return torch.where(hit, texel, self.nowhere)

    def look(self, frame):
        """Marks what ``frame`` shows as seen; returns how many texels each env saw for the first time."""
        texel = self.texels_hit(frame).reshape(-1)
        ray = torch.arange(len(texel), device=texel.device)
        epoch = self.epoch.repeat_interleave(len(texel)//len(self.epoch))                 # of each ray's env
        self.claim[texel] = ray                                                           # one of the rays on a texel wins
        new = (self.claim[texel] == ray) & (self.stamp[texel] != epoch) & (texel != self.nowhere)
        self.stamp[texel] = epoch
        gained = new.view(len(self.epoch), -1).sum(1).float()
        self.count = self.count + gained
        return gained

    def forget(self, envs):
        """Envs marked in the bool mask start over."""
        self.epoch += envs.int()
        self.count = self.count.masked_fill(envs, 0)

    def mask(self):
        """Per texel: has its env seen it since the env's last respawn."""
        return self.stamp[:-1] == self.epoch[self.texel_env]


class Explorer:

    def __init__(self, n_envs, *args, device='cuda', geometries=None, **kwargs):
        if geometries is None:
            geometries = cubicasa.sample(n_envs)
        self.core = core.Core(scene.scenery(geometries, 1, device=device), *args, res=4*64, fov=130, **kwargs)
        c = self.core
        self.device = c.device

        self._mover = modules.MomentumMovement(c)
        self._respawner = modules.RandomSpawns(geometries, c)
        self._rgb = modules.RGB(c, n_agents=1, subsample=4)
        self._depth = modules.Depth(c, n_agents=1, subsample=4)
        self._imu = modules.IMU(c)
        self.action_space = self._mover.space
        self.obs_space = dotdict.dotdict(rgb=self._rgb.space, d=self._depth.space, imu=self._imu.space)

        self._memory = SeenTexels(c.scenery, c.n_envs)
        self._lengths = torch.zeros(c.n_envs, dtype=torch.int, device=c.device)

    # what the tests and `state` look at
    _potential = property(lambda self: self._memory.count)
    _seen = property(lambda self: self._memory.mask())
    _tex_to_env = property(lambda self: self._memory.texel_env)

    def _restart(self, which):
        self._respawner(which.unsqueeze(-1))
        self._memory.forget(which)
        self._lengths.masked_fill_(which, 0)

    def _world(self, reset):
        # pooled RGB-D straight from the render kernel; the reward only needs which texel each ray landed on
        frame = modules.render(self.core, observers=(self._rgb, self._depth), fields=('indices', 'locations'))
        pixels = self.core.res//self._rgb.subsample
        reward = (self._memory.look(frame)/pixels).masked_fill(reset, 0.)      # nothing for the frame after a respawn
        obs = arrdict.arrdict(rgb=self._rgb(frame), d=self._depth(frame), imu=self._imu())
        return arrdict.arrdict(obs=obs, reset=reset, reward=reward)

    @torch.no_grad()
    def reset(self):
        everyone = self.core.env_full(True)
        self._restart(everyone)
        return self._world(everyone)

    @torch.no_grad()
    def step(self, decision):
        self._mover(decision)
        self._lengths += 1
        over = self._lengths >= self._memory.count + EPISODE_SLACK
        self._restart(over)
        return self._world(over)

    def state(self, e=0):
        seen = self._memory.mask()[self._memory.texel_env == e]
        return arrdict.arrdict(core=self.core.state(e), rgb=self._rgb.state(e), d=self._depth.state(e),
                               potential=self._memory.count[e].clone(), seen=seen.clone(),
                               length=self._lengths[e].clone(), max_length=self._memory.count[e].add(EPISODE_SLACK).clone())
