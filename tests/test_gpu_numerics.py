"""The kernels' arithmetic shortcuts against the operations they stand for, bit for bit, on the device:
div_inrange - a correctly rounded binary32 division without the range scaling of the compiler's expansion - over the operand
ranges its call sites guarantee, and sqrt_any over everything."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(n=None, d=None, x=None):
    from megastep_amd import _lib
    h = _lib.lib()
    dev = torch.device('cuda')
    out = {}
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    if n is not None:
        n, d = torch.as_tensor(n, device=dev).contiguous(), torch.as_tensor(d, device=dev).contiguous()
        out['fast'], out['ieee'] = torch.empty_like(n), torch.empty_like(n)
    if x is not None:
        x = torch.as_tensor(x, device=dev).contiguous()
        out['r_any'], out['r_ieee'] = torch.empty_like(x), torch.empty_like(x)
    count = len(n) if n is not None else len(x)
    _lib.check(h.ms_test_arithmetic(ptr(n), ptr(d), ptr(out.get('fast')), ptr(out.get('ieee')), ptr(x), ptr(out.get('r_any')), ptr(out.get('r_ieee')),
                                    count, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}, (None if n is None else (n.cpu().numpy(), d.cpu().numpy()))


def _bits(a):
    return a.view(np.uint32)


def test_division_in_range_is_the_correctly_rounded_quotient():
    """Ten million operand pairs over the ranges the call sites guarantee - divisors 1e-3 .. 1e4 in size (|cross(ray, wall)| of a
    hit, a ray's length, texel weights' sums, a wall's length + 1e-6), numerators 1e-5 .. 1e9 in size, either sign - and the
    awkward ones: numerators a few ulps around k x divisor (quotients next to a rounding boundary), exact quotients, numerators
    at the top and bottom of the range.  Three ways the same bits: div_inrange on the device, the compiler's full division on the
    device, numpy's binary32 division on the host (IEEE round-to-nearest-even)."""
    rng = np.random.RandomState(0)
    m = 10_000_000
    d = (10.**rng.uniform(-3, 4, m)*rng.choice([-1., 1.], m)).astype(np.float32)
    n = (10.**rng.uniform(-5, 9, m)*rng.choice([-1., 1.], m)).astype(np.float32)
    # a million near rounding boundaries: n = q0 x d nudged by a few ulps, q0 a short binary fraction
    k = 1_000_000
    q0 = (rng.randint(1, 1 << 12, k)/float(1 << rng.randint(0, 12))).astype(np.float32)
    near = (q0*d[:k]).astype(np.float32)
    near = (near.view(np.int32) + rng.randint(-3, 4, k).astype(np.int32)).view(np.float32)
    n[:k] = near
    got, _ = _run(n, d)
    want = (n/d).astype(np.float32)                                       # numpy: correctly rounded binary32
    assert np.array_equal(_bits(got['ieee']), _bits(want)), 'the compiler\'s division is not the IEEE quotient?'
    diff = _bits(got['fast']) != _bits(want)
    assert not diff.any(), (int(diff.sum()), n[diff][:5], d[diff][:5], got['fast'][diff][:5], want[diff][:5])
    # zero numerators: zero (the sign of a zero may differ - no call site looks at it)
    z, _ = _run(np.zeros(1000, np.float32), d[:1000].copy())
    assert (z['fast'] == 0).all() and (z['ieee'] == 0).all()


def test_division_in_range_over_the_lighting_sites_whole_declared_range():
    """The dynamic lighting's call site (kernels/lighting.h, `nice`): 2 I / max(d^2, 1) with I in [1e-12, 1e12] and lights and hit
    points within 10^6 m of the origin - numerators 2e-12 .. 2e12, divisors 1 .. 8e12, up to 82 binary orders apart.  ADVICE r5:
    the guard used to let 1e-30 .. 1e30 and 10^15 m through, a range no test covered; it now admits exactly what is swept here."""
    rng = np.random.RandomState(4)
    m = 5_000_000
    n = (2.*10.**rng.uniform(-12, 12, m)).astype(np.float32)
    d = np.maximum(10.**rng.uniform(-2, np.log10(8e12), m), 1.).astype(np.float32)
    # the corners of the box, and quotients next to rounding boundaries at both ends of it
    n[:4], d[:4] = [2e-12, 2e-12, 2e12, 2e12], [1., 8e12, 1., 8e12]
    k = 500_000
    q0 = (rng.randint(1, 1 << 12, k)/float(1 << 11)).astype(np.float32)*(10.**rng.uniform(-24, 12, k)).astype(np.float32)
    near = (q0*d[4:4 + k]).astype(np.float32)
    ok = (near >= 2e-12) & (near <= 2e12)
    near = (near.view(np.int32) + rng.randint(-3, 4, k).astype(np.int32)).view(np.float32)
    n[4:4 + k] = np.where(ok, near, n[4:4 + k])
    got, _ = _run(n, d)
    want = (n/d).astype(np.float32)
    assert np.array_equal(_bits(got['ieee']), _bits(want))
    diff = _bits(got['fast']) != _bits(want)
    assert not diff.any(), (int(diff.sum()), n[diff][:5], d[diff][:5], got['fast'][diff][:5], want[diff][:5])


def test_division_outside_its_range_only_where_the_kernels_say_it_may_differ():
    """Numerators below 2^-104 (the compiler's expansion rescales those): the quotient may differ from the IEEE one there - by a few
    ulps of something below 1e-28, which every call site either throws away (a hit inside the near plane) or writes to a float
    output with a 1e-5 tolerance.  Held to exactly that: tiny numerators give tiny quotients, never anything that could be taken
    for a hit beyond a near plane of centimetres."""
    rng = np.random.RandomState(1)
    m = 1_000_000
    d = (10.**rng.uniform(-3, 4, m)).astype(np.float32)
    n = (10.**rng.uniform(-44, -32, m)).astype(np.float32)                # denormals and just above
    got, _ = _run(n, d)
    assert np.isfinite(got['fast']).all() and (np.abs(got['fast']) < 1e-28).all() and (np.abs(got['ieee']) < 1e-28).all()
    assert np.abs(got['fast'].astype(np.float64) - got['ieee'].astype(np.float64)).max() < 1e-35


def test_sqrt_any_is_sqrtf_everywhere():
    """sqrt_normal for normal arguments, the library's behind a branch for the rest: the same bits as sqrtf for ten million
    arguments over the whole binary32 range, zeros, denormals, infinities and NaNs included."""
    rng = np.random.RandomState(2)
    bits = rng.randint(0, 1 << 31, 10_000_000).astype(np.uint32)          # every non-negative pattern: normals, denormals, inf, NaNs
    x = bits.view(np.float32).copy()
    x[:8] = [0., -0., np.inf, np.nan, 1e-45, 1.17549435e-38, 3.4028235e38, -1.]
    got, _ = _run(x=x)
    both_nan = np.isnan(got['r_any']) & np.isnan(got['r_ieee'])
    same = (_bits(got['r_any']) == _bits(got['r_ieee'])) | both_nan
    assert same.all(), (int((~same).sum()), x[~same][:5], got['r_any'][~same][:5], got['r_ieee'][~same][:5])
    with np.errstate(invalid='ignore'):
        want = np.sqrt(x.astype(np.float32))
    ok = (_bits(got['r_ieee']) == _bits(want)) | (np.isnan(got['r_ieee']) & np.isnan(want))
    assert ok.all()
