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
        self._seen = torch.full_like(self._tex_to_env, False)
        self._potential = self.core.env_full(0.)
        self._lengths = torch.zeros(self.core.n_envs, device=self.core.device, dtype=torch.int)
        self.device = self.core.device
        # bookkeeping for the incremental reward: which ray last claimed a texel, and each env's texel range
        self._claim = torch.full_like(self._tex_to_env, -1)
        line_ends = sc.lines.ends.long()
        self._tex_ends = sc.textures.ends.long()[line_ends - 1]
        self._tex_starts = torch.cat([self._tex_ends.new_zeros(1), self._tex_ends[:-1]])

    def _tex_indices(self, aux):
        sc = self.core.scenery
        mask = aux.indices >= 0
        result = torch.full_like(aux.indices, -1, dtype=torch.long)
        tex_n = (sc.lines.starts[:, None, None, None] + aux.indices)[mask].long()
        tex_w = sc.textures.widths[tex_n].float()
        tex_i = torch.min(torch.floor(tex_w*aux.locations[mask]), tex_w - 1)
        result[mask] = sc.textures.starts[tex_n].long() + tex_i.long()
        return result.unsqueeze(2)

    def _reward(self, r, reset):
        """Reward = newly seen texels per env (reference: explorer.py:45-58). The reference re-counts every texel of
        every env each step (a scatter_add over all of them); here only the texels this step's rays landed on are
        touched: a texel is counted once, by whichever of the rays on it holds the claim, if it was unseen before."""
        tex = self._tex_indices(r).reshape(-1)
        tex = tex[tex >= 0]
        rays = torch.arange(len(tex), device=tex.device)
        self._claim[tex] = rays                              # duplicates: some single ray wins each texel
        fresh = (self._claim[tex] == rays) & ~self._seen[tex]
        self._seen[tex] = True
        potential = self._potential.clone()
        potential.scatter_add_(0, self._tex_to_env[tex], fresh.float())
        reward = (potential - self._potential)/(self.core.res//self._rgb.subsample)
        self._potential = potential
        # Should I render twice so that the last reward is accurate?
        reward[reset] = 0.
        return reward

    def _observe(self, reset):
        # pooled RGB-D straight from the render kernel; the reward only needs which texel each ray landed on
        r = modules.render(self.core, observers=(self._rgb, self._depth), fields=('indices', 'locations'))
        obs = arrdict.arrdict(rgb=self._rgb(r), d=self._depth(r), imu=self._imu())
        return obs, self._reward(r, reset)

    def _reset(self, reset=None):
        self._respawner(reset.unsqueeze(-1))
        envs = reset.nonzero().squeeze(-1)
        if len(envs):                                        # forget what the respawned envs had seen
            starts, lens = self._tex_starts[envs], self._tex_ends[envs] - self._tex_starts[envs]
            offsets = torch.arange(int(lens.sum()), device=envs.device) - torch.repeat_interleave(lens.cumsum(0) - lens, lens)
            self._seen[torch.repeat_interleave(starts, lens) + offsets] = False
        self._potential[reset] = 0
        self._lengths[reset] = 0

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
