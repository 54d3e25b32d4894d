"""``env.step()`` as one HIP graph.

A step of the demo envs is a couple of kernel launches and a handful of small tensor ops with no host synchronisation in
between, which makes the whole of it capturable (``torch.cuda.CUDAGraph``): the host's share of a step - a hundred
microseconds of Python and launch calls against thirty of GPU work for ``Explorer(4096)`` - disappears, and what the kernels
can do is what the caller gets (``bench.py``'s ``env_step`` reports both). No counterpart in the reference, whose step
synchronises with the host (``nonzero``, reference: modules.py:316-320)."""
import torch

from . import arrdict


class GraphedStep:
    """Wraps an env (``reset()``, ``step(decision)`` with ``decision.actions`` an integer tensor): the first ``step`` call
    warms the step up on a side stream, captures it, and from then on every call copies the actions into the captured
    input and replays the graph. The world it returns is the same arrdict of tensors every time, overwritten in place:
    consume (or clone) it before the next step. Everything else is passed through to the env.

    The warm-up is real: the first ``step`` call advances the env ``warmup`` steps under its actions (their rewards and
    resets are not returned) before the captured step runs for the first time - graph capture needs the step's
    allocations to have happened once. Call it before the steps that count, or pass ``warmup=1``."""

    def __init__(self, env, warmup=3):
        self.env = env
        self._warmup = warmup
        self._graph = None
        self._actions = None
        self._world = None

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self):
        return self.env.reset()

    @torch.no_grad()
    def step(self, decision):
        if self._graph is None:
            with torch.cuda.device(self.env.device):                      # (streams and graphs are made on the current device)
                self._actions = decision.actions.clone()
                static = arrdict.arrdict(actions=self._actions)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):                              # (graph capture wants its warm-up elsewhere)
                    for _ in range(max(self._warmup, 1)):
                        self.env.step(static)
                torch.cuda.current_stream().wait_stream(side)
                self._graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph):
                    self._world = self.env.step(static)
        self._actions.copy_(decision.actions)
        self._graph.replay()
        return self._world
