"""A minimal environment: a box, RGB observations, simple movement (reference: megastep/demo/envs/minimal.py:7-31)."""
from ... import modules, core, toys, scene, arrdict, dotdict


class Minimal:

    def __init__(self, n_envs=1, device='cuda'):
        geometries = n_envs*[toys.box()]
        scenery = scene.scenery(geometries, n_agents=1, device=device)
        self.core = core.Core(scenery)
        self.spawner = modules.RandomSpawns(geometries, self.core)
        self.rgb = modules.RGB(self.core)
        self.movement = modules.SimpleMovement(self.core)

        self.obs_space = self.rgb.space
        self.action_space = self.movement.space

    def reset(self):
        self.spawner(self.core.agent_full(True))
        return arrdict.arrdict(obs=self.rgb())

    def step(self, decision):
        self.movement(decision)
        return arrdict.arrdict(obs=self.rgb())

    def state(self, e=0):
        return dotdict.dotdict(core=self.core.state(e), rgb=self.rgb.state(e))
