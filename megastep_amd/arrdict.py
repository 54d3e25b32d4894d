"""dotdicts of arrays/tensors: indexing, assignment and arithmetic are applied to every leaf. Covers the part of
``rebar.arrdict`` (reference: rebar/arrdict.py:11-162) used by core/scene/modules and the envs; written from scratch."""
import operator
import numpy as np
import torch
from . import dotdict as _dd

_BINARY = ['lt', 'le', 'eq', 'ne', 'ge', 'gt', 'add', 'sub', 'mul', 'matmul', 'truediv', 'floordiv', 'mod', 'pow',
           'lshift', 'rshift', 'and', 'or', 'xor']


def _is_key(x):
    return isinstance(x, str) or (isinstance(x, tuple) and len(x) > 0 and all(isinstance(xx, str) for xx in x))


class arrdict(_dd.dotdict):

    def __getitem__(self, x):
        if _is_key(x):
            return super().__getitem__(x)
        return type(self)((k, v[x]) for k, v in self.items())

    def __setitem__(self, x, y):
        if _is_key(x):
            super().__setitem__(x, y)
        elif isinstance(y, dict):
            for k in self:
                self[k][x] = y[k]
        else:
            raise ValueError('Index-assignment into an arrdict needs an arrdict on the right-hand side')

    def __setattr__(self, key, value):
        raise ValueError('Set arrdict entries by key, not by attribute')


def _install():
    def make(name, reflected):
        fn = getattr(operator, f'__{name}__', None) or getattr(operator, f'{name}_')

        def op(self, other):
            if isinstance(other, dict):
                pairs = ((k, (fn(other[k], v) if reflected else fn(v, other[k]))) for k, v in self.items())
            else:
                pairs = ((k, (fn(other, v) if reflected else fn(v, other))) for k, v in self.items())
            return type(self)(pairs)
        return op

    for name in _BINARY:
        setattr(arrdict, f'__{name}__', make(name, False))
        if name not in ('lt', 'le', 'eq', 'ne', 'ge', 'gt'):
            setattr(arrdict, f'__r{name}__', make(name, True))
    arrdict.__hash__ = None


_install()


def _torchify_leaf(a):
    if hasattr(a, 'torchify'):
        return a.torchify()
    a = np.asarray(a)
    if np.issubdtype(a.dtype, np.floating):
        dtype = torch.float32
    elif np.issubdtype(a.dtype, np.integer):
        dtype = torch.int32
    elif a.dtype == np.bool_:
        dtype = torch.bool
    else:
        raise ValueError(f"Can't turn a {a.dtype} array into a tensor")
    return torch.as_tensor(np.array(a), dtype=dtype)


def torchify(tree):
    """numpy -> CPU tensors; floats become float32 and ints int32 (reference: rebar/arrdict.py:66-88)."""
    return _dd.tree_map(_torchify_leaf, tree)


def _numpyify_leaf(t):
    if isinstance(t, tuple):
        return tuple(_numpyify_leaf(x) for x in t)
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy().copy()
    if hasattr(t, 'numpyify'):
        return t.numpyify()
    return t


def numpyify(tree):
    """tensors -> numpy arrays (reference: rebar/arrdict.py:90-100)."""
    return _dd.tree_map(_numpyify_leaf, tree)


def _combine(xs, np_fn, torch_fn, args, kwargs):
    head = xs[0]
    if isinstance(head, dict):
        return type(head)((k, _combine([x[k] for x in xs], np_fn, torch_fn, args, kwargs)) for k in head)
    if isinstance(head, torch.Tensor):
        return torch_fn(list(xs), *args, **kwargs)
    if isinstance(head, np.ndarray):
        return np_fn(list(xs), *args, **kwargs)
    if np.isscalar(head):
        return np.array(list(xs))
    raise ValueError(f"Can't combine {type(head)}")


def stack(xs, *args, **kwargs):
    """Stacks a sequence of arrays / tensors / trees of them (reference: rebar/arrdict.py:102-127)."""
    return _combine(xs, np.stack, torch.stack, args, kwargs)


def cat(xs, *args, **kwargs):
    """Concatenates a sequence of arrays / tensors / trees of them (reference: rebar/arrdict.py:129-153)."""
    return _combine(xs, np.concatenate, torch.cat, args, kwargs)


def _clone_leaf(t):
    if hasattr(t, 'clone'):
        return t.clone()
    if hasattr(t, 'copy'):
        return t.copy()
    return t


def clone(tree):
    return _dd.tree_map(_clone_leaf, tree)
