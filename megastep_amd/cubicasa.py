"""Seeded synthetic stand-in for the Cubicasa5k floorplan sample (reference: megastep/cubicasa.py:177-224).

The reference downloads ~5k real floorplans (network + non-commercial licence); neither is available to this build, and
the benchmark configs call for *synthetic* cubicasa floorplans. :func:`sample` keeps the reference's signature and
return type - a list of geometry dicts ``{id, walls, lights, masks, res}`` - and draws from a deterministic pool of
procedurally generated apartments whose statistics follow the one in-repo datapoint (reference: core.py:103-107:
275 walls + 21 lights in a typical plan):

* axis-aligned apartment, 8-20 m x 6-15 m, recursively split into 10-25 rooms;
* walls are 0.15 m thick and appear as the *outlines* of wall pieces (as the reference's walls are polygon
  exteriors, geometry.py:43-57), broken by 0.9 m door gaps; small pillars pad the count to the target;
* 150-400 wall segments per plan (800-1200 with ``large=True``), all coordinates > MARGIN;
* one light per room at its centroid; masks at 0.2 m with -1 wall / 0 outside / k room.

``oblique=True`` (round 6): the reference's walls are the exteriors of arbitrary SVG polygons (geometry.py:43-57) - nothing
in a real floorplan is aligned with the axes of its coordinate system, and some walls are not aligned with each other. An
oblique plan is the aligned plan of the same index with a few diagonal partitions added (thick pieces cutting room corners
at seeded angles) and then the whole of it - walls, rooms, lights - turned by a seeded angle about its centre and moved
back into the positive quadrant; masks are rasterised from the turned shapes.
"""
import numpy as np
from . import geometry, arrdict

N_UNIQUE = 4992          # the reference dataset's size
THICK = .15
DOOR = .9


def _rect_walls(x0, y0, x1, y1):
    """Outline of an axis-aligned rectangle as 4 oriented segments (counter-clockwise)."""
    c = np.array([[x0, y0], [x1, y0], [x1, y1], [x0, y1]])
    return np.stack([c, np.roll(c, -1, 0)], 1)


def _split_rooms(rng, w, h, n_rooms, min_side):
    rooms = [(0., 0., w, h)]
    for _ in range(10*n_rooms):
        if len(rooms) >= n_rooms:
            break
        areas = np.array([(r[2] - r[0])*(r[3] - r[1]) for r in rooms])
        i = int(rng.choice(len(rooms), p=areas/areas.sum()))
        x0, y0, x1, y1 = rooms[i]
        horizontal = (x1 - x0) < (y1 - y0)          # cut across the long side
        lo, hi = (y0, y1) if horizontal else (x0, x1)
        if hi - lo < 2*min_side:
            continue
        cut = rng.uniform(lo + min_side, hi - min_side)
        rooms.pop(i)
        if horizontal:
            rooms += [(x0, y0, x1, cut), (x0, cut, x1, y1)]
        else:
            rooms += [(x0, y0, cut, y1), (cut, y0, x1, y1)]
    return rooms


def _shared_edges(rooms):
    """(i, j, axis, coord, lo, hi) for every pair of rooms sharing a boundary stretch."""
    out = []
    for i, a in enumerate(rooms):
        for j in range(i + 1, len(rooms)):
            b = rooms[j]
            for axis in (0, 1):                      # axis 0: vertical wall at x = coord
                o = 1 - axis
                for coord_a, coord_b in ((a[2 + axis], b[axis]), (a[axis], b[2 + axis])):
                    if abs(coord_a - coord_b) < 1e-9:
                        lo, hi = max(a[o], b[o]), min(a[2 + o], b[2 + o])
                        if hi - lo > 1e-6:
                            out.append((i, j, axis, coord_a, lo, hi))
    return out


def _find(parent, i):
    while parent[i] != i:
        parent[i] = parent[parent[i]]
        i = parent[i]
    return i


def _turned_piece(cx, cy, length, angle):
    """Outline of a THICK x length wall piece centred at (cx, cy), at `angle` to the x-axis: 4 oriented segments."""
    c, s = np.cos(angle), np.sin(angle)
    u, v = np.array([c, s])*length/2, np.array([-s, c])*THICK/2
    corners = np.array([cx, cy]) + np.array([-u - v, u - v, u + v, -u + v])
    return np.stack([corners, np.roll(corners, -1, 0)], 1)


def floorplan(seed, large=False, oblique=False):
    """One synthetic apartment as a geometry dict (without ``id``)."""
    rng = np.random.RandomState(seed)
    scale = 2.2 if large else 1.
    w, h = scale*rng.uniform(8, 20), scale*rng.uniform(6, 15)
    n_rooms = int(rng.randint(40, 70)) if large else int(rng.randint(10, 26))
    target = int(rng.randint(800, 1201)) if large else int(rng.randint(150, 401))
    rooms = _split_rooms(rng, w, h, n_rooms, 1.7)
    off = geometry.MARGIN + THICK                   # keep every coordinate > MARGIN

    # interior walls: one thick piece per shared edge, split around a door where the rooms need connecting
    pieces = []
    edges = _shared_edges(rooms)
    parent = list(range(len(rooms)))
    for k in rng.permutation(len(edges)):
        i, j, axis, coord, lo, hi = edges[k]
        ri, rj = _find(parent, i), _find(parent, j)
        door = (hi - lo > DOOR + .6) and (ri != rj or rng.uniform() < .25)
        spans = [(lo, hi)]
        if door:
            parent[ri] = rj
            d0 = rng.uniform(lo + .3, hi - DOOR - .3)
            spans = [(lo, d0), (d0 + DOOR, hi)]
        for s0, s1 in spans:
            if s1 - s0 < 1e-3:
                continue
            if axis == 0:
                pieces.append((coord - THICK/2, s0, coord + THICK/2, s1))
            else:
                pieces.append((s0, coord - THICK/2, s1, coord + THICK/2))
    # exterior shell: four thick slabs
    pieces += [(-THICK, -THICK, w + THICK, 0.), (-THICK, h, w + THICK, h + THICK),
               (-THICK, 0., 0., h), (w, 0., w + THICK, h)]

    walls = [_rect_walls(*p) for p in pieces]
    # pillars / ducts against room corners pad the segment count up to the target
    n_pillars = max((target - 4*len(walls)) // 4, 0)
    for _ in range(n_pillars):
        x0, y0, x1, y1 = rooms[int(rng.randint(len(rooms)))]
        s = rng.uniform(.15, .45)
        px = rng.choice([x0 + THICK/2, x1 - THICK/2 - s])
        py = rng.uniform(y0 + THICK/2, max(y1 - THICK/2 - s, y0 + THICK/2 + 1e-3))
        walls.append(_rect_walls(px, py, px + s, py + s))
    spaces = [np.array([[x0, y0], [x1, y0], [x1, y1], [x0, y1]]) for x0, y0, x1, y1 in rooms]
    if oblique:
        # (a stream of its own: the aligned plan of this index is what it was)
        orng = np.random.RandomState((seed + 0x9E3779B1) % (2**32))
        # diagonal partitions: a piece across a corner of a room, 0.6-1.6 m from it, at 25-65 degrees to the room's walls
        for i in orng.permutation(len(rooms))[:max(len(rooms)//4, 2)]:
            x0, y0, x1, y1 = rooms[i]
            d = orng.uniform(.6, min(1.6, .45*min(x1 - x0, y1 - y0)))
            sx, sy = orng.choice([-1, 1]), orng.choice([-1, 1])
            corner = np.array([x0 if sx > 0 else x1, y0 if sy > 0 else y1])
            slope = np.radians(orng.uniform(25, 65))
            # from (d / tan) along one wall to d along the other
            p, q = corner + [sx*d/np.tan(slope), 0.], corner + [0., sy*d]
            mid, vec = (p + q)/2, q - p
            walls.append(_turned_piece(mid[0], mid[1], max(np.hypot(*vec) - 2*THICK, .2), np.arctan2(vec[1], vec[0])))
        walls = np.concatenate(walls)
        theta = orng.uniform(0, 2*np.pi)
        c, s = np.cos(theta), np.sin(theta)
        rot = np.array([[c, s], [-s, c]])                       # row vectors: p @ rot turns p by theta
        centre = np.array([w/2, h/2])
        walls = (walls - centre) @ rot
        spaces = [(sp - centre) @ rot for sp in spaces]
        shift = off - walls.reshape(-1, 2).min(0)               # back into the positive quadrant, every coordinate > MARGIN
        walls = walls + shift
        spaces = [sp + shift for sp in spaces]
    else:
        walls = np.concatenate(walls) + off
        spaces = [sp + off for sp in spaces]
    return arrdict.arrdict(
        walls=walls,
        lights=geometry.centroids(spaces),
        masks=geometry.masks(walls, spaces),
        res=geometry.RES)


_cache = {}


def _key(i, large, oblique=False):
    """A plan's cache key: (index, large) for the aligned plans - as it has always been - and (index, large, True) for the oblique."""
    return (int(i), bool(large), True) if oblique else (int(i), bool(large))


def _name(key):
    return f'synthetic-{"L" if key[1] else "S"}{"o" if len(key) > 2 else ""}{key[0]:04d}'


def _build(key):
    return floorplan(1000003*int(key[1]) + key[0], key[1], len(key) > 2)


def _plan(i, large, oblique=False):
    key = _key(i, large, oblique)
    if key not in _cache:
        _cache[key] = arrdict.arrdict(id=_name(key), **_build(key))
    return _cache[key]


def _make(keys):
    return [(key, _build(key)) for key in keys]


def prefetch(indices, large=False, workers=None, context='fork', oblique=False):
    """Generates the plans ``indices`` that are not cached yet on ``workers`` worker processes (a plan takes ~7 ms of
    numpy on one core, a large one 23; the benchmark's Explorer-style worlds want thousands of distinct ones). ``context``
    'fork' (cheap; call it before the process has touched its GPU) or 'subprocess' (fresh numpy-only interpreters that
    load this file and geometry.py on their own: safe at any time - from a test that has been using the GPU for a minute -
    for a fraction of a second of start-up). Same plans as the lazy path - a plan is a function of its index alone."""
    import multiprocessing as mp
    import os
    todo = sorted({_key(i, large, oblique) for i in indices} - set(_cache))
    workers = min(workers or (os.cpu_count() or 1), 32, max(len(todo)//16, 1))
    if workers <= 1 or len(todo) < 64:
        return
    if context == 'subprocess':
        return _prefetch_subprocess(todo, workers)
    with mp.get_context(context).Pool(workers) as pool:
        chunks = [todo[i:i + 8] for i in range(0, len(todo), 8)]
        results = pool.imap_unordered(_make, chunks)
        try:
            for _ in chunks:
                for key, plan in results.next(timeout=60):           # (a pool that stalls is abandoned: the lazy path makes the rest)
                    _cache[key] = arrdict.arrdict(id=_name(key), **plan)
        except mp.TimeoutError:
            pool.terminate()


_WORKER = """
import importlib.util, pickle, sys, types
here, todo_path, out_path = sys.argv[1:4]
pkg = types.ModuleType('_ms_plans'); pkg.__path__ = [here]; sys.modules['_ms_plans'] = pkg
stub = types.ModuleType('_ms_plans.arrdict'); stub.arrdict = dict; sys.modules['_ms_plans.arrdict'] = stub     # (no torch in here)
for name in ('geometry', 'cubicasa'):
    spec = importlib.util.spec_from_file_location('_ms_plans.' + name, here + '/' + name + '.py')
    mod = importlib.util.module_from_spec(spec); sys.modules['_ms_plans.' + name] = mod; spec.loader.exec_module(mod)
keys = pickle.load(open(todo_path, 'rb'))
pickle.dump([(k, dict(mod._build(k))) for k in keys], open(out_path, 'wb'), protocol=4)
"""


def _prefetch_subprocess(todo, workers):
    """The missing plans made by `workers` fresh interpreters (numpy only), exchanged through pickles in a temp dir."""
    import os
    import pickle
    import subprocess
    import sys
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for w in range(workers):
            part = todo[w::workers]
            if not part:
                continue
            src, dst = os.path.join(tmp, f'todo{w}.pkl'), os.path.join(tmp, f'plans{w}.pkl')
            pickle.dump(part, open(src, 'wb'))
            procs.append((subprocess.Popen([sys.executable, '-c', _WORKER, here, src, dst], env=dict(os.environ, OMP_NUM_THREADS='1')), dst))
        for proc, dst in procs:
            try:
                if proc.wait(timeout=300) == 0:
                    for key, plan in pickle.load(open(dst, 'rb')):
                        _cache[key] = arrdict.arrdict(id=_name(key), **plan)
            except subprocess.TimeoutExpired:                          # (the lazy path makes what is missing)
                proc.kill()


def save_cache(path):
    """The plans generated so far, pickled (numpy arrays only): a profiled run, which must not fork workers, loads them."""
    import pickle
    with open(path, 'wb') as f:
        pickle.dump({k: dict(v) for k, v in _cache.items()}, f, protocol=4)


def load_cache(path):
    import pickle
    with open(path, 'rb') as f:
        for k, v in pickle.load(f).items():
            _cache.setdefault(k, arrdict.arrdict(**v))


def sample(n_geometries, split='training', seed=1, large=False, n_unique=N_UNIQUE, workers=0, context='fork', oblique=False):
    """A deterministic sample of ``n_geometries`` floorplans; same arguments, same sample (reference:
    cubicasa.py:177-224). ``split`` is 90/10 ``training``/``test`` or ``all`` over ``n_unique`` plans; plans are
    generated lazily and repeat cyclically when more are asked for than the split holds. ``large`` and ``n_unique`` are
    extensions for the big-map benchmark point and for cheap tests; ``workers`` > 1 generates the missing plans on that
    many worker processes first (see :func:`prefetch`; ``context``: how they are started); ``oblique``: the plans turned by
    seeded angles, with diagonal partitions (see the module's docstring)."""
    cutoff = int(.9*n_unique)
    order = np.random.RandomState(seed).permutation(n_unique)
    if split == 'training':
        order = order[:cutoff]
    elif split == 'test':
        order = order[cutoff:]
    elif split != 'all':
        raise ValueError('Split must be train/test/all')
    if workers and workers > 1:
        prefetch(order[:n_geometries], large, workers, context, oblique)
    return [_plan(order[i % len(order)], large, oblique) for i in range(n_geometries)]
