"""Multi-agent deathmatch: agents score by keeping an opponent in the middle of their view
(reference: megastep/demo/envs/deathmatch.py:9-115)."""
import numpy as np
import torch
from ... import modules, core, spaces, scene, cubicasa, arrdict, dotdict

CLEARANCE = 1.


@dotdict.mapping
def expand(x):
    B, A = x.shape[:2]
    return x.reshape(B*A, 1, *x.shape[2:])


@dotdict.mapping
def collapse(x, n_agents):
    B = x.shape[0]
    return x.reshape(B//n_agents, n_agents, *x.shape[2:])


class Deathmatch:

    def __init__(self, n_envs, n_agents, *args, device='cuda', geometries=None, **kwargs):
        geometries = cubicasa.sample(max(n_envs//4, 1)) if geometries is None else geometries
        scenery = scene.scenery(geometries, n_agents, device=device)
        self.core = core.Core(scenery, *args, res=4*128, fov=70, **kwargs)
        self._rgb = modules.RGB(self.core, n_agents=1, subsample=4)
        self._depth = modules.Depth(self.core, n_agents=1, subsample=4)
        self._imu = modules.IMU(self.core, n_agents=1)
        self._movement = modules.MomentumMovement(self.core, n_agents=1)
        self._spawner = modules.RandomSpawns(geometries, self.core)

        self.action_space = self._movement.space
        self.obs_space = dotdict.dotdict(
            rgb=self._rgb.space, d=self._depth.space, imu=self._imu.space, health=spaces.MultiVector(1, 1))

        bounds = np.stack([np.array(g['masks'].shape)*g['res'] for g in geometries])
        self._bounds = arrdict.torchify(bounds).to(self.core.device)
        self._health = self.core.agent_full(np.nan)
        self._damage = self.core.agent_full(np.nan)

        self.n_envs = self.core.n_envs*self.core.n_agents
        self.device = self.core.device

    def _reset(self, reset=None):
        reset = (self._health <= 0) if reset is None else reset
        self._spawner(reset)
        self._health.masked_fill_(reset, 1.)
        self._damage.masked_fill_(reset, 0.)
        return reset.reshape(-1)

    def _shoot(self, opponents):
        res = opponents.size(-1)
        middle = slice(res//2 - 1, res//2 + 1)
        agents = torch.arange(self.core.n_agents, device=self.core.device)
        matchings = (opponents[:, :, None] == agents[None, None, :, None, None])[..., middle].any(-1).any(-1)
        self.matchings = matchings

        hits = matchings.sum(2).float()
        wounds = matchings.sum(1).float()
        self._damage[:] += .05*hits

        pos = self.core.agents.positions
        outside = (pos < -CLEARANCE).any(-1) | (pos > (self._bounds[:, None] + CLEARANCE)).any(-1)
        # 5% damage per hit, .1% damage per timestep
        self._health[:] += -.05*(wounds + outside) - .001
        return hits.reshape(-1)

    def _observe(self):
        # pooled RGB-D straight from the render kernel; shooting only needs the line each ray landed on
        r = modules.render(self.core, observers=(self._rgb, self._depth), fields=('indices',))
        line_idxs = modules.downsample(r.indices, self._rgb.subsample)[..., self._rgb.subsample//2]
        obj_idxs = torch.div(line_idxs, len(self.core.scenery.model), rounding_mode='floor')
        mask = (0 <= line_idxs) & (obj_idxs < self.core.n_agents)
        opponents = obj_idxs.where(mask, torch.full_like(line_idxs, -1))
        hits = self._shoot(opponents)
        obs = arrdict.arrdict(
            rgb=self._rgb(r), d=self._depth(r), imu=self._imu(), health=self._health.unsqueeze(-1).clone())
        return obs, hits

    @torch.no_grad()
    def reset(self):
        reset = self._reset(self.core.agent_full(True))
        obs, reward = self._observe()
        return arrdict.arrdict(obs=expand(obs), reward=reward, reset=reset)

    @torch.no_grad()
    def step(self, decision):
        reset = self._reset()
        self._movement(collapse(decision, self.core.n_agents))
        obs, reward = self._observe()
        return arrdict.arrdict(obs=expand(obs), reward=reward, reset=reset)

    def state(self, e=0):
        return arrdict.arrdict(
            core=self.core.state(e), rgb=self._rgb.state(e), d=self._depth.state(e),
            health=self._health[e].clone(), damage=self._damage[e].clone(),
            matchings=self.matchings[e].clone(), bounds=self._bounds[e].clone())
