"""Example environments built on the core (reference: megastep/demo/envs/). The RL agent/training loop of the
reference's demo package is learner-side and out of scope; these envs are the callers of the hot path."""
from .envs import Minimal, Explorer, Deathmatch  # noqa: F401
