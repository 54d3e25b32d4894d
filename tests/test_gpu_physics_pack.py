"""physics_kernel's PACK - several envs side by side in one wave (large worlds of few agents per env) - changes who does the
work, never the result: every packing against one env per wave bit for bit, the plain step, the step with the movement
prologue and with the envs' bookkeeping, envs the lists do not cover next to envs they do, and against the oracle."""
import numpy as np
import pytest
import torch

from tests import util
from tests.test_gpu_wallgrid import _world

pytestmark = pytest.mark.gpu


@pytest.fixture
def pack():
    from megastep_amd import _lib
    h = _lib.lib()
    yield h.ms_debug_physics_pack
    h.ms_debug_physics_pack(0)


def _bits(t):
    t = t.detach().cpu()
    return t.numpy().view(np.int32) if t.dtype == torch.float32 else t.numpy()


def _state(c):
    a = c.agents
    return [a.positions.clone(), a.angles.clone(), a.velocity.clone(), a.angvelocity.clone()]


def _restore(c, s):
    a = c.agents
    a.positions[:] = s[0]; a.angles[:] = s[1]; a.velocity[:] = s[2]; a.angvelocity[:] = s[3]


@pytest.mark.parametrize('n_envs,n_agents,packs', [(37, 1, (2, 8, 16, 64)), (50, 2, (3, 8, 32)), (21, 4, (2, 5, 16)), (9, 8, (2, 8)), (5, 12, (2, 5))])
def test_envs_side_by_side_in_one_wave_step_like_envs_alone(pack, n_envs, n_agents, packs):
    from megastep_amd import cuda
    c, _ = _world(n_envs, n_agents, seed=4)
    rng = np.random.RandomState(9)
    ref = util.OracleWorld(c)
    for step, speed in enumerate((3., 3., 40., 1.)):
        util.random_velocities(c, rng, speed=speed)
        if step == 1:                                                    # envs the lists do not cover, among envs they do
            pos = c.agents.positions.clone()
            pos[1, 0] = torch.tensor([-30., 2.]); pos[n_envs - 1, n_agents - 1] = float('nan'); pos[3, 0] = torch.tensor([1e4, 1e4])
            c.agents.positions[:] = pos
            c.agents.velocity[4] *= 20.                                  # faster than the lists reach
            c.agents.velocity[2] *= 1e-7                                 # crawling
        before = _state(c)
        ref.pull_agents(c)
        pack(1)
        p1 = cuda.physics(c.scenery, c.agents)
        want = [p1.progress.clone()] + _state(c)
        prog_ref, agents_ref = ref.physics()
        if step == 0:                                                    # (the odd poses of step 1 on: compared among the kernels only)
            util.assert_physics_matches(c, p1, prog_ref, agents_ref)
        for k in packs:
            _restore(c, before)
            pack(k)
            pk = cuda.physics(c.scenery, c.agents)
            for x, y in zip([pk.progress] + _state(c), want):
                assert np.array_equal(_bits(x), _bits(y)), (k, step)
    assert (want[0] < 1).any()


def test_the_movement_prologue_and_the_bookkeeping_in_packed_waves(pack):
    """ms_step_physics with everything on - actions -> velocities, lifespans, respawns before the step, the IMU reading - for
    envs side by side against envs alone."""
    from megastep_amd import cuda
    N, A, S = 45, 2, 7
    c, _ = _world(N, A, seed=6)
    g = torch.Generator(device='cuda').manual_seed(3)
    table = torch.tensor([[0., 0., 0.], [.3, 0., 0.], [-.3, 0., 0.], [0., .3, 0.], [0., -.3, 0.], [0., 0., 40.], [0., 0., -40.]], device='cuda')
    spawn_p = c.agents.positions[:, :, None, :].repeat(1, 1, S, 1).contiguous() + .05*torch.rand((N, A, S, 2), device='cuda', generator=g)
    spawn_a = 360*torch.rand((N, A, S), device='cuda', generator=g) - 180
    outs = {}
    start = _state(c)
    for k in (1, 4, 16, 32):
        _restore(c, start)
        pack(k)
        gk = torch.Generator(device='cuda').manual_seed(11)
        ages = torch.zeros((N, A), dtype=torch.int32, device='cuda')
        maxima = torch.randint(2, 6, (N, A), dtype=torch.int32, device='cuda', generator=gk)
        rows = []
        for step in range(8):
            actions = torch.randint(0, 7, (N, A), device='cuda', generator=gk)
            mask = torch.rand((N, A), device='cuda', generator=gk) < .1
            choices = torch.randint(0, S, (N, A), device='cuda', generator=gk)
            fresh = torch.randint(2, 6, (N, A), dtype=torch.int32, device='cuda', generator=gk)
            imu = torch.zeros((N, A, 3), device='cuda')
            p = cuda.physics(c.scenery, c.agents, movement=(actions, table, .875),
                             respawn=dict(mask=mask, choices=choices, positions=spawn_p, angles=spawn_a, after=bool(step % 2)),
                             lifespans=dict(lifespans=ages, max_lifespans=maxima, fresh=fresh), imu=(imu, 360., 10.))
            rows.append([p.progress.clone(), imu, mask.clone(), ages.clone(), maxima.clone()] + _state(c))
        outs[k] = rows
    for k in (4, 16, 32):
        for step, (got, want) in enumerate(zip(outs[k], outs[1])):
            for x, y in zip(got, want):
                assert np.array_equal(_bits(x), _bits(y)), (k, step)


def test_the_packing_ms_step_physics_picks_for_a_large_world(pack):
    """From 3072 envs up with a wall grid the library packs by itself (8192 envs of one agent: two to a wave): same bits as
    one env per wave, and the renderer's heading cache with them."""
    from megastep_amd import cuda
    c, _ = _world(8192, 1, seed=2, n_unique=64)
    rng = np.random.RandomState(1)
    util.random_velocities(c, rng, speed=3.)
    start = _state(c)
    got = {}
    for k in (1, 0):
        _restore(c, start)
        pack(k)
        p = cuda.physics(c.scenery, c.agents)
        r = cuda.render(c.scenery, c.agents, fields=('distances', 'indices'))
        got[k] = [p.progress.clone(), r.distances.clone(), r.indices.clone()] + _state(c)
    for x, y in zip(got[0], got[1]):
        assert np.array_equal(_bits(x), _bits(y))
