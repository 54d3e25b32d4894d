"""dotdict / arrdict behaviours the host code relies on (reference: rebar/dotdict.py, rebar/arrdict.py, docs/concepts.rst)."""
import numpy as np
import pytest
import torch
from megastep_amd import arrdict, dotdict


def test_dot_access_and_leaf_forwarding():
    d = dotdict.dotdict(a=np.zeros((2, 3)), b=dotdict.dotdict(c=np.zeros((2, 4))))
    assert d.a.shape == (2, 3) and d.b.c.shape == (2, 4)
    shapes = d.shape
    assert shapes.a == (2, 3) and shapes.b.c == (2, 4)
    summed = d.sum()
    assert summed.a == 0 and summed.b.c == 0
    with pytest.raises(AttributeError):
        dotdict.dotdict(a=1).nonexistent
    assert 'a' in dir(d)


def test_map_starmap_mapping_leaves():
    d = dotdict.dotdict(a=1, b=dotdict.dotdict(c=2))
    assert d.map(lambda x, k: x + k, 10) == {'a': 11, 'b': {'c': 12}}
    assert d.starmap(lambda x, y: x*y, d) == {'a': 1, 'b': {'c': 4}}
    assert dotdict.mapping(lambda x: -x)(d) == {'a': -1, 'b': {'c': -2}}
    assert dotdict.mapping(lambda x: -x)(5) == -5
    assert dotdict.starmapping(lambda x, y: x - y)(d, d) == {'a': 0, 'b': {'c': 0}}
    assert dotdict.leaves(d) == [1, 2]
    assert isinstance(d.map(lambda x: x).b, dotdict.dotdict)
    assert 'a' in str(d) and 'dotdict' in str(d)


def test_arrdict_indexing_and_assignment():
    d = arrdict.arrdict(x=torch.arange(6.).reshape(3, 2), y=arrdict.arrdict(z=torch.arange(3)))
    assert d[1].x.tolist() == [2., 3.] and d[1].y.z.item() == 1
    assert d[[0, 2]].y.z.tolist() == [0, 2]
    d[[0, 2]] = arrdict.arrdict(x=torch.zeros(2, 2), y=arrdict.arrdict(z=torch.tensor([7, 8])))
    assert d.x.sum().item() == 5. and d.y.z.tolist() == [7, 1, 8]
    d['w'] = torch.ones(3)
    assert d.w.shape == (3,)
    with pytest.raises(ValueError):
        d.w = 1
    with pytest.raises(ValueError):
        d[0] = 3


def test_arrdict_arithmetic_and_conversions():
    d = arrdict.arrdict(a=np.arange(3.), b=arrdict.arrdict(c=np.ones((3, 2))))
    assert ((d + d).a == 2*np.arange(3.)).all() and ((2*d).b.c == 2).all() and ((1 - d).a == 1 - np.arange(3.)).all()
    assert ((d > .5).a == [False, True, True]).all()
    t = arrdict.torchify(d)
    assert t.a.dtype == torch.float32 and t.b.c.shape == (3, 2)
    assert arrdict.torchify(np.arange(3)).dtype == torch.int32 and arrdict.torchify(np.array([True])).dtype == torch.bool
    back = arrdict.numpyify(t)
    assert isinstance(back.a, np.ndarray) and isinstance(back, arrdict.arrdict)
    assert arrdict.cat([d, d]).a.shape == (6,) and arrdict.stack([d, d]).b.c.shape == (2, 3, 2)
    assert arrdict.stack([1., 2.]).tolist() == [1., 2.]
    c = arrdict.clone(t)
    c.a[0] = 5
    assert t.a[0] == 0
    assert t.to('cpu').a.device.type == 'cpu'
