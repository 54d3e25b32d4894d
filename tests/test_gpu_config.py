"""Every Core carries its own constants (`cuda.initialize` kept one process-global config, as kernels.cu:12-27 does,
and two Cores of different res / fov in one process silently rendered with the last one's)."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def test_two_cores_of_different_shapes_step_alternately():
    from megastep_amd import core, cubicasa, cuda, modules, scene
    np.random.seed(3)
    geometries = cubicasa.sample(5, n_unique=16, seed=4)
    worlds = []
    for res, fov, fps, n_agents in ((64, 130., 10., 2), (128, 70., 20., 3), (48, 100., 10., 1)):
        scenery = scene.scenery(geometries, n_agents, device='cuda', random=np.random.RandomState(res))
        c = core.Core(scenery, res=res, fov=fov, fps=fps)
        util.spawn(c, geometries, seed=res)
        worlds.append((c, util.OracleWorld(c)))
    # the process-global fallback now holds the LAST core's constants; every core must still be served with its own
    assert cuda._config.res == 48
    rng = np.random.RandomState(0)
    for step in range(3):
        for c, ref in worlds if step % 2 == 0 else worlds[::-1]:
            util.random_velocities(c, rng)
            ref.pull_agents(c)
            p = cuda.physics(c.scenery, c.agents)                         # the drop-in two-argument calls
            r = cuda.render(c.scenery, c.agents)
            assert r.distances.shape == (c.n_envs, c.n_agents, c.res)
            util.assert_physics_matches(c, p, *ref.physics())
            util.assert_render_matches(c, r, ref.render())
            # ... and through the modules, which go by the core
            obs = modules.render(c)
            assert obs.screen.shape[-1] == c.res
    # an explicit config wins over the agents' own
    c, ref = worlds[0]
    wide = cuda.config(c.agent_radius, 2*c.res, c.fov, c.fps)
    assert cuda.render(c.scenery, c.agents, config=wide).distances.shape[-1] == 2*c.res
    with pytest.raises(RuntimeError):
        cuda.render(c.scenery, c.agents, config=(c.agent_radius, 64, 130., 10.))
    torch.cuda.synchronize()
