"""Single-agent exploration: reward for every texel seen for the first time
(reference: megastep/demo/envs/explorer.py:8-115)."""
import torch
from ... import modules, core, scene, cubicasa, arrdict, dotdict


class Explorer:

    def __init__(self, n_envs, *args, device='cuda', geometries=None, **kwargs):
        geometries = cubicasa.sample(n_envs) if geometries is None else geometries
        scenery = scene.scenery(geometries, 1, device=device)
        self.core = core.Core(scenery, *args, res=4*64, fov=130, **kwargs)
        self._rgb = modules.RGB(self.core, n_agents=1, subsample=4)
        self._depth = modules.Depth(self.core, n_agents=1, subsample=4)
        self._mover = modules.MomentumMovement(self.core)
        self._imu = modules.IMU(self.core)
        self._respawner = modules.RandomSpawns(geometries, self.core)

        self.action_space = self._mover.space
        self.obs_space = dotdict.dotdict(rgb=self._rgb.space, d=self._depth.space, imu=self._imu.space)

        sc = self.core.scenery
        self._tex_to_env = sc.lines.inverse[sc.textures.inverse.long()].long()
        self._potential = self.core.env_full(0.)
        self._lengths = torch.zeros(self.core.n_envs, device=self.core.device, dtype=torch.int)
        self.device = self.core.device
        # Bookkeeping for the incremental reward, laid out so that a step needs neither a pass over every texel nor
        # a host sync: a texel counts as seen while its stamp equals its env's epoch (a respawn bumps the epoch and
        # thereby forgets the env's texels); `_claim` records which ray last landed on a texel. One extra slot at the
        # end of both arrays takes the rays that hit nothing; it always reads as seen.
        n_tex = len(self._tex_to_env)
        self._epoch = torch.ones(self.core.n_envs, device=self.device, dtype=torch.int32)
        self._stamp = torch.zeros(n_tex + 1, device=self.device, dtype=torch.int32)
        self._claim = torch.full((n_tex + 1,), -1, device=self.device, dtype=torch.long)
        self._trash = torch.tensor(n_tex, device=self.device)

    @property
    def _seen(self):
        """Per texel: has its env seen it since the env's last respawn."""
        return self._stamp[:-1] == self._epoch[self._tex_to_env]

    def _tex_indices(self, aux):
        """The texel every ray landed on, (n_env, n_agent, 1, res); the trash slot for rays that hit nothing."""
        sc = self.core.scenery
        valid = aux.indices >= 0
        line = (sc.lines.starts[:, None, None, None] + aux.indices.clamp(min=0)).long()
        tex_w = sc.textures.widths[line].float()
        tex_i = torch.min(torch.floor(tex_w*aux.locations), tex_w - 1)
        tex = sc.textures.starts[line].long() + torch.where(valid, tex_i, torch.zeros_like(tex_i)).long()
        return torch.where(valid, tex, self._trash)

    def _reward(self, r, reset):
        """Reward = newly seen texels per env (reference: explorer.py:45-58). The reference re-counts every texel of
        every env each step (a scatter_add over all of them); here only the texels this step's rays landed on are
        touched: a texel is counted once, by whichever of the rays on it holds the claim, if it was unseen before."""
        tex = self._tex_indices(r).reshape(-1)
        rays = torch.arange(len(tex), device=tex.device)
        epoch = self._epoch.repeat_interleave(len(tex)//self.core.n_envs)     # of the env each ray belongs to
        self._claim[tex] = rays                              # duplicates: some single ray wins each texel
        fresh = (self._claim[tex] == rays) & (self._stamp[tex] != epoch) & (tex != self._trash)
        self._stamp[tex] = epoch
        potential = self._potential + fresh.view(self.core.n_envs, -1).sum(1).float()
        reward = (potential - self._potential)/(self.core.res//self._rgb.subsample)
        self._potential = potential
        # Should I render twice so that the last reward is accurate?
        reward = reward.masked_fill(reset, 0.)
        return reward

    def _observe(self, reset):
        # pooled RGB-D straight from the render kernel; the reward only needs which texel each ray landed on
        r = modules.render(self.core, observers=(self._rgb, self._depth), fields=('indices', 'locations'))
        obs = arrdict.arrdict(rgb=self._rgb(r), d=self._depth(r), imu=self._imu())
        return obs, self._reward(r, reset)

    def _reset(self, reset=None):
        self._respawner(reset.unsqueeze(-1))
        self._epoch += reset.int()                           # forget what the respawned envs had seen
        self._potential = self._potential.masked_fill(reset, 0)
        self._lengths.masked_fill_(reset, 0)

    @torch.no_grad()
    def reset(self):
        reset = self.core.env_full(True)
        self._reset(reset)
        obs, reward = self._observe(reset)
        return arrdict.arrdict(obs=obs, reset=reset, reward=reward)

    @torch.no_grad()
    def step(self, decision):
        self._mover(decision)
        self._lengths += 1
        reset = (self._lengths >= self._potential + 200)
        self._reset(reset)
        obs, reward = self._observe(reset)
        return arrdict.arrdict(obs=obs, reset=reset, reward=reward)

    def state(self, e=0):
        return arrdict.arrdict(
            core=self.core.state(e), rgb=self._rgb.state(e), d=self._depth.state(e),
            potential=self._potential[e].clone(), seen=self._seen[self._tex_to_env == e].clone(),
            length=self._lengths[e].clone(), max_length=self._potential[e].add(200).clone())
