"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on identical seeded inputs.

Bar (BASELINE.json north_star): bit-exact collision masks and hit indices; <= 1e-5 on float positions / pixels."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _world(n_envs, n_agents, res, fov, seed=0, large=False, toy=None):
    from megastep_amd import core, cubicasa, scene, toys
    np.random.seed(seed)
    if toy is not None:
        geometries = n_envs*[getattr(toys, toy)()]
    else:
        geometries = cubicasa.sample(n_envs, n_unique=64, seed=seed + 1, large=large)
    scenery = scene.scenery(geometries, n_agents, device='cuda', random=np.random.RandomState(seed))
    c = core.Core(scenery, res=res, fov=fov, fps=10)
    util.spawn(c, geometries, seed=seed)
    return c, geometries


@pytest.mark.parametrize('n_envs,n_agents,res,fov,toy', [
    (1, 1, 8, 130, 'box'),
    (3, 1, 64, 130, 'box'),
    (2, 2, 64, 70, 'column'),
    (8, 1, 64, 130, None),
    (8, 4, 64, 130, None),
    (5, 4, 128, 70, None),
    (3, 3, 100, 90, None),      # ragged ray group (100 = 64 + 36), A does not divide the wave count
    (2, 6, 32, 130, None),      # more agents than waves per workgroup
    (2, 4, 512, 70, None),      # the reference's own Deathmatch resolution
])
def test_step_matches_oracle(n_envs, n_agents, res, fov, toy):
    from megastep_amd import cuda
    c, geometries = _world(n_envs, n_agents, res, fov, toy=toy)
    ref = util.OracleWorld(c)
    np.testing.assert_allclose(c.scenery.baked.vals.cpu().numpy(), ref.bake(), rtol=0, atol=1e-5)
    ref.pull_baked(c)
    rng = np.random.RandomState(7)
    exact = []
    for step in range(4):
        util.random_velocities(c, rng, speed=4. if step % 2 else 40.)
        ref.pull_agents(c)
        p = cuda.physics(c.scenery, c.agents)
        r = cuda.render(c.scenery, c.agents)
        prog_ref, agents_ref = ref.physics()
        render_ref = ref.render()
        util.assert_physics_matches(c, p, prog_ref, agents_ref)
        util.assert_render_matches(c, r, render_ref)
        np.testing.assert_allclose(c.scenery.lines.vals.cpu().numpy(), ref.scene.lines_vals, rtol=0, atol=1e-6)
        exact.append([util.exact_fraction(p.progress.cpu().numpy(), prog_ref),
                      util.exact_fraction(r.distances.cpu().numpy(), render_ref['distances']),
                      util.exact_fraction(r.screen.cpu().numpy(), render_ref['screen'])])
    print('bitwise-equal fractions (progress, distances, screen):', np.mean(exact, 0))


def test_agents_see_each_other():
    """Two agents facing each other in a box: exercises agent-line hits and the dynamic lighting path."""
    from megastep_amd import cuda
    c, _ = _world(4, 2, 64, 70, toy='box')
    c.agents.positions[:] = torch.tensor([[2.5, 3.5], [4.5, 3.5]], device=c.device)
    c.agents.angles[:] = torch.tensor([0., 180.], device=c.device)
    ref = util.OracleWorld(c)
    ref.bake(); ref.pull_baked(c); ref.pull_agents(c)
    r = cuda.render(c.scenery, c.agents)
    want = ref.render()
    idx = r.indices.cpu().numpy()
    assert ((idx >= 0) & (idx < 16)).any(), 'expected rays to land on the other agent'
    assert (r.screen.cpu().numpy()[(idx >= 0) & (idx < 16)] > 0).any()
    util.assert_render_matches(c, r, want)


def test_known_answer_from_the_docs():
    """reference: docs/tutorials/minimal-env/index.rst:140-145 - box(5), agent at (3, 3), velocity (1000, 0)."""
    from megastep_amd import cuda
    c, _ = _world(128, 1, 64, 130, toy='box')
    c.agents.positions[:] = torch.as_tensor([3., 3.], device=c.device)
    c.agents.angles[:] = 0.
    c.agents.velocity[:] = torch.as_tensor([1000., 0.], device=c.device)
    p = cuda.physics(c.scenery, c.agents)
    np.testing.assert_allclose(c.agents.positions.cpu().numpy(), np.tile([5.8649, 3.0], (128, 1, 1)), atol=5e-5)
    assert (p.progress.cpu().numpy() < 1).all()
    assert (c.agents.velocity.cpu().numpy() == 0).all()


def test_errors_are_loud():
    from megastep_amd import cuda, core, scene, toys
    scenery = scene.scenery([toys.box()], 1, device='cpu', bake=False)
    with pytest.raises(RuntimeError, match='GPU'):
        cuda.bake(scenery)
    agents = core._init_agents(1, 1, 'cpu')
    with pytest.raises(RuntimeError, match='GPU'):
        cuda.physics(scenery, agents)
