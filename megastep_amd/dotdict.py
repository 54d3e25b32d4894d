"""Attribute-access dictionaries: the subset of ``rebar.dotdict`` (reference: rebar/dotdict.py:7-177) that the
simulation host code leans on - dot access, leaf forwarding, tree ``map``/``starmap`` - written from scratch."""
import functools


def _is_tree(x):
    return isinstance(x, dict)


def _rebuild(template, pairs):
    return type(template)(pairs)


def tree_map(f, tree, *rest, args=(), kwargs=None):
    """Applies ``f`` (a callable, or the name of a method) leaf-wise over ``tree`` and the parallel trees in ``rest``."""
    kwargs = kwargs or {}
    if _is_tree(tree):
        return _rebuild(tree, [(k, tree_map(f, v, *(r[k] for r in rest), args=args, kwargs=kwargs)) for k, v in tree.items()])
    if isinstance(f, str):
        return getattr(tree, f)(*rest, *args, **kwargs)
    return f(tree, *rest, *args, **kwargs)


def mapping(f):
    """Lifts ``f(leaf, *args, **kwargs)`` to trees of dicts (reference: rebar/dotdict.py:123-149)."""
    @functools.wraps(f) if callable(f) else (lambda g: g)
    def lifted(x, *args, **kwargs):
        return tree_map(f, x, args=args, kwargs=kwargs)
    return lifted


def starmapping(f):
    """Lifts ``f(leaf0, leaf1, ...)`` to parallel trees of dicts (reference: rebar/dotdict.py:151-171)."""
    @functools.wraps(f) if callable(f) else (lambda g: g)
    def lifted(x, *others):
        return tree_map(f, x, *others)
    return lifted


def leaves(tree):
    """The leaves of a tree of dicts, depth first."""
    if _is_tree(tree):
        return [leaf for v in tree.values() for leaf in leaves(v)]
    return [tree]


def _describe(v, width):
    if isinstance(v, dotdict):
        return str(v)
    if isinstance(v, (list, set, dict)):
        return f'{type(v).__name__}({len(v)},)'
    if hasattr(v, 'shape') and hasattr(v, 'dtype'):
        return f'{type(v).__name__}({tuple(v.shape)}, {v.dtype})'
    if hasattr(v, 'shape'):
        return f'{type(v).__name__}({tuple(v.shape)})'
    text = str(v).splitlines() or ['']
    return text[0][:width] + (' ...' if len(text) > 1 or len(text[0]) > width else '')


class dotdict(dict):
    """A dict whose keys are also attributes. Asking for an attribute that is not a key asks every value for it
    instead and returns a dotdict of the answers, so ``d.shape``, ``d.cuda()``, ``d.float()`` work on whole trees."""

    def __getattr__(self, key):
        if key.startswith('__'):
            raise AttributeError(key)
        try:
            return self[key]
        except KeyError:
            pass
        try:
            return type(self)((k, getattr(v, key)) for k, v in self.items())
        except AttributeError:
            raise AttributeError(f"No key '{key}', and not every leaf has an attribute '{key}'") from None

    def __call__(self, *args, **kwargs):
        return type(self)((k, v(*args, **kwargs)) for k, v in self.items())

    def __dir__(self):
        return sorted(set(super().__dir__()) | {k for k in self if isinstance(k, str)})

    def __str__(self):
        pad = 4 + max([len(str(k)) for k in self] + [0])
        out = [f'{type(self).__name__}:']
        for k, v in self.items():
            first, *more = _describe(v, 119 - pad).splitlines() or ['']
            out.append(f'{str(k):<{pad}}{first}')
            out.extend(' '*pad + line for line in more)
        return '\n'.join(out)

    __repr__ = __str__

    def copy(self):
        return type(self)(self)

    def pipe(self, f, *args, **kwargs):
        return f(self, *args, **kwargs)

    def map(self, f, *args, **kwargs):
        return tree_map(f, self, args=args, kwargs=kwargs)

    def starmap(self, f, *others):
        return tree_map(f, self, *others)
