"""The state holder around the kernels (reference: megastep/core.py:10-150)."""
import numpy as np
import torch
from . import cuda, arrdict, dotdict

AGENT_WIDTH = .15
TEXTURE_RES = .05

#: collision radius and near camera plane
AGENT_RADIUS = 1/2**.5*AGENT_WIDTH


def gamma_encode(x):
    """Linear RGB -> viewable values."""
    return x**(1/2.2)


def gamma_decode(x):
    """Viewable RGB -> linear (interpolatable) values."""
    return x**2.2


def _init_agents(n_envs, n_agents, device='cuda', config=None):
    """A zeroed :class:`~megastep_amd.cuda.Agents` (reference: core.py:24-31)."""
    zeros = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=device)
    return cuda.Agents(angles=zeros(n_envs, n_agents), positions=zeros(n_envs, n_agents, 2),
                       angvelocity=zeros(n_envs, n_agents), velocity=zeros(n_envs, n_agents, 2), config=config)


class Core:

    def __init__(self, scenery, res=64, fov=130, fps=10):
        """The rendering and physics interface (reference: core.py:35-89).

        Holds ``scenery`` and a zeroed ``agents``; the tensors hanging off it are the state of the world, advanced by
        :func:`megastep_amd.cuda.physics` and observed with :func:`megastep_amd.cuda.render`.

        Unlike the reference, ``res`` is not capped at 1024: rays are processed in 64-wide groups, one wavefront each.
        """
        self.n_envs = len(scenery.lines.widths)
        self.n_agents = scenery.n_agents
        self.res = res
        self.fov = fov
        self.agent_radius = AGENT_RADIUS
        self.fps = fps
        self.random = np.random.RandomState(1)
        self.device = scenery.model.device

        assert fov < 180, 'FOV should be less than 180°'

        # The reference keeps these four in process-global device constants (kernels.cu:12-27): one Core's res / fov per
        # process. Here each Core keeps its own (`config`, passed by value with every launch) and hangs it on its agents,
        # which is where cuda.physics / cuda.render look first - several Cores of different shapes step side by side.
        # initialize() is still called, for callers of the drop-in two-argument cuda.render(scenery, agents) on Agents
        # they built themselves.
        self.config = cuda.config(self.agent_radius, self.res, self.fov, self.fps)
        cuda.initialize(self.agent_radius, self.res, self.fov, self.fps)
        self.scenery = scenery
        self.agents = _init_agents(self.n_envs, self.n_agents, self.device, self.config)
        self.progress = torch.ones((self.n_envs, self.n_agents), device=self.device)

    def state(self, e):
        """A dotdict tree of the state of env ``e`` (reference: core.py:91-121)."""
        options = {k: getattr(self, k) for k in ('n_envs', 'n_agents', 'res', 'fov', 'agent_radius', 'fps')}
        return arrdict.clone(dotdict.dotdict(
            **options, scenery=self.scenery.state(e), agents=self.agents.state(e), progress=self.progress[e]))

    _DTYPES = {bool: torch.bool, int: torch.int32, float: torch.float32}

    def env_full(self, x):
        """An (n_envs,) tensor full of ``x`` on the core's device (reference: core.py:136-142)."""
        return torch.full((self.n_envs,), x, device=self.device, dtype=self._DTYPES[type(x)])

    def agent_full(self, x):
        """An (n_envs, n_agents) tensor full of ``x`` on the core's device (reference: core.py:144-150)."""
        return torch.full((self.n_envs, self.n_agents), x, device=self.device, dtype=self._DTYPES[type(x)])
