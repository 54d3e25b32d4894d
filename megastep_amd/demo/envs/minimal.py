"""The smallest useful environment: every env is the 5 m toy box with one agent in it, the observation is what the
agent sees (RGB, full resolution), the seven movement actions apply at once (no momentum). It is the skeleton of the
reference's tutorial (docs/tutorials/minimal-env, megastep/demo/envs/minimal.py:7-31): spawn - observe on reset,
move - observe on step. The observation comes pooled (subsample 1) straight from the render kernel."""
import torch

from ... import arrdict, core, dotdict, modules, scene, toys


class Minimal:

    def __init__(self, n_envs=1, device='cuda'):
        box = toys.box()
        rooms = [box for _ in range(n_envs)]
        self.core = core.Core(scene.scenery(rooms, n_agents=1, device=device))
        self.device = self.core.device

        self.rgb = modules.RGB(self.core)
        self.movement = modules.SimpleMovement(self.core)
        self.spawner = modules.RandomSpawns(rooms, self.core)
        self.obs_space, self.action_space = self.rgb.space, self.movement.space

    def _world(self):
        frame = modules.render(self.core, observers=(self.rgb,), fields=())
        return arrdict.arrdict(obs=self.rgb(frame))

    @torch.no_grad()
    def reset(self):
        everyone = self.core.agent_full(True)
        self.spawner(everyone)
        return self._world()

    @torch.no_grad()
    def step(self, decision):
        if self.core.device.type == 'cuda':
            # the velocities, the physics step and the render: ONE launch - an agent of 64 rays is a single wavefront (cuda.step_render)
            frame = modules.move_render(self.core, self.movement, decision, observers=(self.rgb,), fields=())
            return arrdict.arrdict(obs=self.rgb(frame))
        self.movement(decision)                  # sets the velocities and runs physics, one launch
        return self._world()

    def state(self, e=0):
        return dotdict.dotdict(rgb=self.rgb.state(e), core=self.core.state(e))
