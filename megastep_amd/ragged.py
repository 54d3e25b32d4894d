"""Ragged arrays: arrays of arrays backed by one contiguous array (reference: megastep/ragged.py:7-75).

``Ragged(vals, widths)`` gives a :class:`RaggedNumpy` for numpy inputs and a ``cuda.Ragged{1,2,3}D`` for tensors."""
import numbers
import numpy as np
from . import arrdict, cuda


class RaggedNumpy:
    """A ragged backed by numpy arrays. ``starts``/``ends`` index ``vals``; ``inverse`` maps each row of ``vals`` to
    the sub-array that owns it."""

    def __init__(self, vals, widths):
        widths = np.asarray(widths)
        if widths.sum() != vals.shape[0]:
            raise ValueError(f'widths sum to {widths.sum()} but vals has {vals.shape[0]} rows')
        self.vals, self.widths = vals, widths
        self.ends = widths.cumsum().astype(int)
        self.starts = self.ends - widths
        self.inverse = np.repeat(np.arange(len(widths)), widths).astype(int)

    def __len__(self):
        return len(self.widths)

    def __getitem__(self, x):
        if isinstance(x, numbers.Integral):
            return self.vals[self.starts[x]:self.ends[x]]
        if isinstance(x, slice):
            start, stop, step = x.indices(len(self.widths))
            if step != 1:
                raise ValueError('Ragged slices must have step 1')
            if stop <= start:
                return RaggedNumpy(self.vals[:0], self.widths[:0])
            return RaggedNumpy(self.vals[self.starts[start]:self.ends[stop - 1]], self.widths[start:stop])
        raise ValueError(f'Can\'t handle index "{x}"')

    def torchify(self):
        return Ragged(arrdict.torchify(self.vals), arrdict.torchify(self.widths))

    def __repr__(self):
        return f'{type(self).__name__}({self.widths})'


def Ragged(vals, widths):
    """Numpy in -> :class:`RaggedNumpy`; tensors in -> ``cuda.Ragged{ndim}D`` (reference: ragged.py:56-75)."""
    if isinstance(vals, np.ndarray):
        return RaggedNumpy(vals, widths)
    if vals.ndim not in (1, 2, 3):
        raise RuntimeError(f'Ragged tensors must be 1-3 dimensional, not {vals.ndim}')
    return getattr(cuda, f'Ragged{vals.ndim}D')(vals, widths)
