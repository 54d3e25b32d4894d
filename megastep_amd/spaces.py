"""Shape holders for observation/action spaces (reference: megastep/spaces.py:1-28)."""


class MultiEmpty:
    pass


class _Shaped:
    def __init__(self, *shape):
        self.shape = tuple(shape)

    def __repr__(self):
        return f'{type(self).__name__}{self.shape}'


class MultiVector(_Shaped):
    def __init__(self, n_agents, dim):
        super().__init__(n_agents, dim)


class MultiImage(_Shaped):
    def __init__(self, n_agents, C, H, W):
        super().__init__(n_agents, C, H, W)


class MultiConstant(_Shaped):
    def __init__(self, n_agents):
        super().__init__(n_agents)


class MultiDiscrete(_Shaped):
    def __init__(self, n_agents, n_actions):
        super().__init__(n_agents, n_actions)
