"""What an env's observations and actions look like, per agent-row: nothing but shapes, which is all a policy
constructor needs (same five names and ``.shape`` tuples as megastep/spaces.py:1-28). The leading axis is the
agents-per-row axis the modules are built with."""


class MultiEmpty:
    """No observation (or no action) at all."""
    shape = ()


class MultiVector:
    """A float vector of ``dim`` entries per agent (IMU, health)."""

    def __init__(self, n_agents, dim):
        self.shape = (n_agents, dim)


class MultiImage:
    """A ``C`` x ``H`` x ``W`` float image per agent (RGB, depth)."""

    def __init__(self, n_agents, C, H, W):
        self.shape = (n_agents, C, H, W)


class MultiConstant:
    """One fixed value per agent: a placeholder action."""

    def __init__(self, n_agents):
        self.shape = (n_agents,)


class MultiDiscrete:
    """One of ``n_actions`` choices per agent (movement)."""

    def __init__(self, n_agents, n_actions):
        self.shape = (n_agents, n_actions)
