"""Geometry helpers (reference: megastep/geometry.py:9-137).

A *geometry* is a dict with ``walls`` (W, 2, 2), ``lights`` (I, 2), ``masks`` (H, W) int16 and ``res``. The reference
rasterises masks with rasterio + shapely (``all_touched=True``); those are not dependencies here, so :func:`masks`
carries its own conservative rasteriser with the same cell conventions (-1 wall, 0 free, k >= 1 room k).
SVG parsing of the cubicasa dataset is offline data preparation and out of scope.
"""
import numpy as np

MARGIN = 1.
RES = .2
SCALE = 100


def cyclic_pairs(xs):
    """Pairs ``(xs[i], xs[i+1])``, wrapping the last one round to the start (reference: geometry.py:15-18)."""
    xs = list(xs)
    return [(xs[i], xs[(i + 1) % len(xs)]) for i in range(len(xs))]


def signed_area(points):
    return sum(x[0]*y[1] - x[1]*y[0] for x, y in cyclic_pairs(points))


def unique(walls):
    """Drops walls that duplicate an earlier wall in either direction, to 1mm (reference: geometry.py:35-41)."""
    fwd = ((walls[:, None] - walls[None, :])**2).sum(-1).sum(-1)**.5
    bwd = ((walls[:, None] - walls[None, :, ::-1])**2).sum(-1).sum(-1)**.5
    dup = np.tril((fwd < 1e-3) | (bwd < 1e-3), -1)
    return walls[~dup.any(1)]


def mask_shape(*pointsets):
    """(H, W) of the mask that covers all the points plus MARGIN (reference: geometry.py:74-79)."""
    points = np.concatenate([np.asarray(p).reshape(-1, 2) for p in pointsets])
    assert points.min() > 0, 'Masker currently requires the points to be in the top-right quadrant'
    r, t = points.max(0) + MARGIN
    return int(t/RES) + 1, int(r/RES) + 1


def _cells_touching_segment(a, b, pad, shape, res):
    """Boolean (H, W) sub-block + its offset: cells whose square, grown by ``pad``, the segment a-b crosses."""
    H, W = shape
    lo, hi = np.minimum(a, b) - pad, np.maximum(a, b) + pad
    j0, j1 = max(int(np.floor(lo[0]/res)), 0), min(int(np.floor(hi[0]/res)), W - 1)
    y_top = H*res
    i0, i1 = max(int(np.floor((y_top - hi[1])/res)), 0), min(int(np.floor((y_top - lo[1])/res)), H - 1)
    if j1 < j0 or i1 < i0:
        return None, (0, 0)
    d = b - a
    if abs(d[0]) < 1e-12 or abs(d[1]) < 1e-12:
        # An axis-aligned segment (every wall of the synthetic floorplans): the slab test below separates into a test per
        # column and a test per row - the same arithmetic on two short vectors instead of two (rows x columns) arrays.
        j, i = np.arange(j0, j1 + 1), np.arange(i0, i1 + 1)
        okx = _slab(a[0], d[0], j*res - pad, (j + 1)*res + pad)
        oky = _slab(a[1], d[1], y_top - (i + 1)*res - pad, y_top - i*res + pad)
        return oky[:, None] & okx[None, :], (i0, j0)
    jj, ii = np.meshgrid(np.arange(j0, j1 + 1), np.arange(i0, i1 + 1))
    xmin, xmax = jj*res - pad, (jj + 1)*res + pad
    ymin, ymax = y_top - (ii + 1)*res - pad, y_top - ii*res + pad
    # slab clipping of the segment against each grown cell
    t0, t1 = np.zeros(jj.shape), np.ones(jj.shape)
    ok = np.ones(jj.shape, bool)
    for p0, dd, mn, mx in ((a[0], d[0], xmin, xmax), (a[1], d[1], ymin, ymax)):
        if abs(dd) < 1e-12:
            ok &= (p0 >= mn) & (p0 <= mx)
        else:
            ta, tb = (mn - p0)/dd, (mx - p0)/dd
            t0 = np.maximum(t0, np.minimum(ta, tb))
            t1 = np.minimum(t1, np.maximum(ta, tb))
    return ok & (t0 <= t1), (i0, j0)


def _slab(p0, dd, mn, mx):
    """One axis of the slab test for a segment that does not move along the other: the cells (intervals mn..mx) it meets."""
    if abs(dd) < 1e-12:
        return (p0 >= mn) & (p0 <= mx)
    ta, tb = (mn - p0)/dd, (mx - p0)/dd
    return np.maximum(0., np.minimum(ta, tb)) <= np.minimum(1., np.maximum(ta, tb))


def _inside(poly, x, y):
    """Even-odd point-in-polygon for arrays of points."""
    inside = np.zeros(x.shape, bool)
    for (x0, y0), (x1, y1) in cyclic_pairs(poly):
        if y0 == y1:
            continue
        crosses = ((y0 <= y) & (y < y1)) | ((y1 <= y) & (y < y0))
        xi = x0 + (y - y0)*(x1 - x0)/(y1 - y0)
        inside ^= crosses & (x < xi)
    return inside


def masks(walls, spaces, res=RES):
    """A masking array from (W, 2, 2) walls and a list of room polygons: 1, 2, ... for the rooms, 0 for free space and
    -1 for walls (reference: geometry.py:81-93). Cell (i, j) covers x in [j, j+1]*res, y in [H-i-1, H-i]*res, and any
    cell a shape touches is marked."""
    walls = np.asarray(walls, dtype=float)
    H, W = shape = mask_shape(walls, *spaces)
    out = np.zeros(shape, dtype=np.int16)
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    cx, cy = res*(jj + .5), res*(H - ii - .5)
    for k, poly in enumerate(spaces):
        poly = np.asarray(poly, dtype=float)
        touched = _inside(poly, cx, cy)
        for a, b in cyclic_pairs(poly):
            blk, (i0, j0) = _cells_touching_segment(np.asarray(a), np.asarray(b), 0., shape, res)
            if blk is not None:
                touched[i0:i0 + blk.shape[0], j0:j0 + blk.shape[1]] |= blk
        out[touched] = k + 1
    d = walls[:, 1] - walls[:, 0]
    aligned = (np.abs(d) < 1e-12).any(1)
    ii, jj = _cells_touching_aligned(walls[aligned], .01, shape, res)       # (all of a synthetic floorplan's walls)
    out[ii, jj] = -1
    for a, b in walls[~aligned]:
        blk, (i0, j0) = _cells_touching_segment(a, b, .01, shape, res)
        if blk is not None:
            out[i0:i0 + blk.shape[0], j0:j0 + blk.shape[1]][blk] = -1
    return out


def _ragged_arange(first, count):
    """(owner, value): for every k the values first[k] .. first[k] + count[k] - 1, laid end to end."""
    owner = np.repeat(np.arange(len(count)), count)
    return owner, np.arange(len(owner)) - np.repeat(np.cumsum(count) - count, count) + first[owner]


def _cells_touching_aligned(walls, pad, shape, res):
    """:func:`_cells_touching_segment` for many axis-aligned segments at once - the same tests, cell for cell, as flat
    (rows, columns) of the touched cells (a floorplan's few hundred walls cost a few hundred tiny numpy calls each otherwise)."""
    H, W = shape
    if len(walls) == 0:
        return np.zeros(0, int), np.zeros(0, int)
    a, b = walls[:, 0], walls[:, 1]
    d = b - a
    lo, hi = np.minimum(a, b) - pad, np.maximum(a, b) + pad
    y_top = H*res
    j0 = np.maximum(np.floor(lo[:, 0]/res).astype(int), 0)
    j1 = np.minimum(np.floor(hi[:, 0]/res).astype(int), W - 1)
    i0 = np.maximum(np.floor((y_top - hi[:, 1])/res).astype(int), 0)
    i1 = np.minimum(np.floor((y_top - lo[:, 1])/res).astype(int), H - 1)
    nj, ni = np.maximum(j1 - j0 + 1, 0), np.maximum(i1 - i0 + 1, 0)
    nj, ni = np.where(ni > 0, nj, 0), np.where(nj > 0, ni, 0)

    def slab(p0, dd, mn, mx):                       # _slab, with the still-or-moving choice made per element
        still = np.abs(dd) < 1e-12
        safe = np.where(still, 1., dd)
        ta, tb = (mn - p0)/safe, (mx - p0)/safe
        return np.where(still, (p0 >= mn) & (p0 <= mx), np.maximum(0., np.minimum(ta, tb)) <= np.minimum(1., np.maximum(ta, tb)))

    wj, j = _ragged_arange(j0, nj)
    okx = slab(a[wj, 0], d[wj, 0], j*res - pad, (j + 1)*res + pad)
    wi, i = _ragged_arange(i0, ni)
    oky = slab(a[wi, 1], d[wi, 1], y_top - (i + 1)*res - pad, y_top - i*res + pad)
    # the cells of a wall: its touched rows x its touched columns
    wj, j, wi, i = wj[okx], j[okx], wi[oky], i[oky]
    cols_of = np.bincount(wj, minlength=len(walls))
    col_first = np.cumsum(cols_of) - cols_of
    per_row = cols_of[wi]                                                   # every touched row meets all its wall's touched columns
    row, k = _ragged_arange(np.zeros(len(wi), int), per_row)
    return i[row], j[col_first[wi[row]] + k]


def centroids(spaces):
    """Area centroids of the room polygons, (n, 2) (reference: geometry.py:95-97)."""
    out = []
    for poly in spaces:
        p = np.asarray(poly, dtype=float)
        q = np.roll(p, -1, 0)
        w = p[:, 0]*q[:, 1] - q[:, 0]*p[:, 1]
        area = w.sum()/2
        out.append(((p + q)*w[:, None]).sum(0)/(6*area) if abs(area) > 1e-12 else p.mean(0))
    return np.array(out).reshape(-1, 2)


def centers(indices, shape, res):
    """Mask (i, j) indices -> (x, y) of the cell centres (reference: geometry.py:110-122)."""
    i, j = indices[..., 0] + .5, indices[..., 1] + .5
    return res*np.stack([j, shape[0] - i], -1)


def indices(coords, shape, res):
    """(x, y) coordinates -> (i, j) indices of the containing cell (reference: geometry.py:124-137)."""
    x, y = coords[..., 0], coords[..., 1]
    i = (shape[0] - y/res).clip(0, shape[0] - 1)
    j = (x/res).clip(0, shape[1] - 1)
    return np.stack([i, j], -1).astype(int)
