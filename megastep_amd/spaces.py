"""What an env's observations and actions look like, per agent-row: nothing but shapes, which is all a policy
constructor needs. Same five names and ``.shape`` tuples as megastep/spaces.py:1-28; the envs and observation modules
of this package build them, user code reads ``space.shape``.

A space is the tuple ``(n_agents, *per_agent_dims)``. The leading axis is the agents-per-row axis the modules are built
with (the demo envs always use 1: they flatten agents into rows)."""
import operator


def _space(name, fields, doc):
    """Makes a space class whose constructor takes ``fields`` (positive ints) and whose shape is those, in order."""

    def __init__(self, *dims, **named):
        given = dict(zip(fields, dims), **named)
        if len(dims) > len(fields) or set(given) != set(fields):
            raise TypeError(f'{name}({", ".join(fields)}): got {dims or ""}{named or ""}')
        try:
            shape = tuple(operator.index(given[f]) for f in fields)          # ints of any flavour, nothing else
        except TypeError:
            shape = (0,)
        if min(shape) < 1:
            raise ValueError(f'{name}: every one of {fields} must be a positive int, got {given}')
        self.shape = shape

    def __repr__(self):
        return f'{name}({", ".join(f"{f}={v}" for f, v in zip(fields, self.shape))})'

    def __eq__(self, other):
        return type(other) is type(self) and other.shape == self.shape

    def __hash__(self):
        return hash((name, self.shape))

    members = dict(__init__=__init__, __repr__=__repr__, __eq__=__eq__, __hash__=__hash__, __doc__=doc,
                   n_agents=property(lambda self: self.shape[0], doc='agents per row'))
    return type(name, (), members)


class MultiEmpty:
    """No observation (or no action) at all."""

    shape = ()

    def __repr__(self):
        return 'MultiEmpty()'


MultiVector = _space('MultiVector', ('n_agents', 'dim'), 'A float vector of ``dim`` entries per agent (IMU, health).')
MultiImage = _space('MultiImage', ('n_agents', 'C', 'H', 'W'), 'A ``C`` x ``H`` x ``W`` float image per agent (RGB, depth).')
MultiConstant = _space('MultiConstant', ('n_agents',), 'One fixed value per agent: a placeholder action.')
MultiDiscrete = _space('MultiDiscrete', ('n_agents', 'n_actions'), 'One of ``n_actions`` choices per agent (movement).')
