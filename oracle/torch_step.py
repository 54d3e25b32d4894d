"""A pure-PyTorch CPU step - physics + render - of the reference's algorithm (TEST INFRASTRUCTURE / BASELINE).

BASELINE.json's north_star asks for "a pure-PyTorch CPU step on the box's own host cores" as the reported-only
baseline next to the GPU numbers. The reference itself has no CPU path (docs/faq.rst:23-26), so this is a port: the
same arithmetic as ``megastep_oracle.c`` (which restates megastep/src/kernels.cu:36-475 statement by statement),
vectorised over envs, agents and rays with torch tensor ops on padded ``(n_envs, max_lines)`` arrays; the
order-dependent nearest-hit fold (kernels.cu:352-377) runs as a Python loop over the line index. Like everything
under ``oracle/`` it may only be used by ``tests/`` (where it is checked against the C oracle) and by ``bench.py``'s
``cpu_baseline`` leg; the product never imports it.
"""
import ctypes
import numpy as np
import torch

_libm = ctypes.CDLL('libm.so.6')
_libm.tanf.restype = ctypes.c_float
_libm.tanf.argtypes = [ctypes.c_float]

F = np.float32
AMBIENT, LUMINANCE = F(.1), F(2.)


def _cross(a, b):
    return a[..., 0]*b[..., 1] - a[..., 1]*b[..., 0]


def _dot(a, b):
    return a[..., 0]*b[..., 0] + a[..., 1]*b[..., 1]


def _len(a):
    return torch.sqrt(a[..., 0]*a[..., 0] + a[..., 1]*a[..., 1])


def _intersect(P, U, Q, V):
    """kernels.cu:67-89; (inf, inf) for near-parallel lines."""
    UxV = _cross(U, V)
    PQ = Q - P
    inf = torch.full_like(UxV, float('inf'))
    par = UxV.abs() < 1.e-3
    return torch.where(par, inf, _cross(PQ, V)/UxV), torch.where(par, inf, _cross(PQ, U)/UxV)


def _project(P, U, Q):
    """kernels.cu:91-107."""
    u = _len(U) + 1e-6
    PQ = Q - P
    return _dot(PQ, U)/(u*u), _cross(PQ, U).abs()/u


def _sensibilize(p):
    """kernels.cu:109-118: clamp(.99 p, 0, 1), NaN -> 0."""
    q = p*.99
    return torch.where(q > 0, torch.where(q < 1, q, torch.ones_like(q)), torch.zeros_like(q))


def _sincospi(x):
    """megastep_oracle.c `oracle_sincospi`: exact binary32 range reduction, binary64 Taylor kernel rounded once."""
    y = x - 2.*torch.round(x*0.5)
    nq = torch.round(2.*y)
    z = y - 0.5*nq
    q = nq.to(torch.int64) & 3
    zd = z.double()
    w = zd*zd
    ps = torch.full_like(w, -2.1915353447830217e-05)
    for c in (0.00046630280576761255, -0.0073704309457143504, 0.08214588661112823, -0.5992645293207921,
              2.5501640398773455, -5.16771278004997, 3.141592653589793):
        ps = ps*w + c
    ps = ps*zd
    pc = torch.full_like(w, 4.303069587032947e-06)
    for c in (-0.0001046381049248457, 0.0019295743094039231, -0.02580689139001406, 0.2353306303588932,
              -1.3352627688545895, 4.0587121264167685, -4.934802200544679, 1.0):
        pc = pc*w + c
    S, C = ps.float(), pc.float()
    s = torch.where(q == 0, S, torch.where(q == 1, C, torch.where(q == 2, -S, -C)))
    c = torch.where(q == 0, C, torch.where(q == 1, -S, torch.where(q == 2, -C, S)))
    return s, c


def _remainder(a, b):
    m = torch.fmod(a, b)
    return torch.where((m != 0) & (m < 0), m + b, m)          # b > 0 here


def _normalize_degrees(a):
    return _remainder(_remainder(a, 360.) + 180., 360.) - 180.


class World:
    """Padded torch copies of an oracle scene dict (see oracle.py) for the vectorised step."""

    def __init__(self, scene, agent_radius, res, fov, fps):
        t = lambda a, dt=None: torch.as_tensor(np.ascontiguousarray(a), dtype=dt)
        self.A, self.M = int(scene['n_agents']), scene['model'].shape[0]
        self.model = t(scene['model'], torch.float32)
        lw = t(scene['lines_widths'], torch.int64)
        iw = t(scene['lights_widths'], torch.int64)
        self.N = len(lw)
        self.lines_starts = lw.cumsum(0) - lw
        lights_starts = iw.cumsum(0) - iw
        Lmax, Imax = int(lw.max()), max(int(iw.max()), 1)
        ar = torch.arange(Lmax)
        self.lmask = ar[None] < lw[:, None]
        src = (self.lines_starts[:, None] + ar[None]).clamp(max=max(int(lw.sum()) - 1, 0))
        self.lines = t(scene['lines_vals'], torch.float32).reshape(-1, 4)[src]*self.lmask[..., None]     # (N, Lmax, 4)
        ai = torch.arange(Imax)
        self.imask = ai[None] < iw[:, None]
        isrc = (lights_starts[:, None] + ai[None]).clamp(max=max(int(iw.sum()) - 1, 0))
        lights = t(scene['lights_vals'], torch.float32).reshape(-1, 3)
        self.lights = (lights[isrc] if len(lights) else torch.zeros(self.N, Imax, 3))*self.imask[..., None]
        self.tex = t(scene['textures_vals'], torch.float32).reshape(-1, 3)
        self.tex_widths = t(scene['textures_widths'], torch.int64)
        self.tex_starts = self.tex_widths.cumsum(0) - self.tex_widths
        self.baked = t(scene['baked_vals'], torch.float32)
        self.R, self.radius, self.fps = int(res), F(agent_radius), F(fps)
        self.half_screen = F(_libm.tanf(F(F(F(3.14159265358979323846)/F(180.))*F(fov))/2.))

    # ---- physics (kernels.cu:119-230) ---------------------------------------------------------------------------
    def physics(self, agents):
        """agents: dict of torch tensors, updated in place; returns progress (N, A)."""
        A, AF, R_ = self.A, self.A*self.M, self.radius
        p, v = agents['positions'], agents['velocity']/self.fps
        x = torch.ones(p.shape[:2])
        # agent-agent (kernels.cu:119-133,193-200)
        r2 = F(F(1.001)*F(2.))*R_
        p0, p1 = p[:, :, None], p[:, None, :]
        dv = v[:, :, None] - v[:, None, :]
        s, d = _project(p0, dv, p1)
        backoff = torch.sqrt(r2*r2 - d*d)/_len(dv)
        cc = torch.where((0 < s) & (d < r2), _sensibilize(s - backoff), torch.ones_like(s))
        cc = torch.where(torch.eye(A, dtype=torch.bool)[None], torch.ones_like(cc), cc)
        x = torch.minimum(x, cc.amin(2))
        # agent-wall (kernels.cu:135-171,202-206)
        r = F(1.001)*R_
        walls, wmask = self.lines[:, AF:], self.lmask[:, AF:]
        if walls.shape[1]:
            la, lb = walls[:, None, :, :2], walls[:, None, :, 2:]
            lv = lb - la
            P, V = p[:, :, None], v[:, :, None]
            one = torch.ones(P.shape[0], A, walls.shape[1])
            ms, mt = _intersect(P, V, la, lv)
            dp = _project(la, lv, P)[1]
            xs = torch.where((0 < ms) & (ms < 1) & (0 < mt) & (mt < 1), _sensibilize((1 - r/dp)*ms), one)
            vlen = _len(V)
            for end in (la, lb):
                es, ed = _project(P, V, end)
                xs = torch.minimum(xs, torch.where((0 < es) & (ed < r), _sensibilize(es - torch.sqrt(r*r - ed*ed)/vlen), one))
            ss, sd = _project(la, lv, P + V)
            xs = torch.minimum(xs, torch.where((0 < ss) & (ss < 1) & (sd < r), _sensibilize((dp - r)/(dp - sd)), one))
            xs = torch.where(wmask[:, None], xs, one)
            x = torch.minimum(x, xs.amin(2))
        # epilogue (kernels.cu:224-227)
        agents['positions'] += x[..., None]*agents['velocity']/self.fps
        agents['velocity'][x < 1] = 0.
        agents['angles'][:] = _normalize_degrees(agents['angles'] + x*agents['angvelocity']/self.fps)
        agents['angvelocity'][x < 1] = 0.
        return x

    # ---- render (kernels.cu:297-475) ----------------------------------------------------------------------------
    def _light_intensity(self, C, env):
        """kernels.cu:238-268 for points C (D, 2) of envs `env` (D,)."""
        AF = self.A*self.M
        acc = torch.full((len(C),), float(AMBIENT))
        walls, wmask = self.lines[env][:, AF:], self.lmask[env][:, AF:]            # (D, Lw, 4)
        la, lv = walls[..., :2], walls[..., 2:] - walls[..., :2]
        for i in range(self.lights.shape[1]):
            I = self.lights[env, i]                                               # (D, 3)
            s, t = _intersect(I[:, None, :2], (C - I[:, :2])[:, None], la, lv)
            blocked = ((t > 0) & (t < 1) & (s > 0) & (s < .999) & wmask).any(1)
            dx, dy = I[:, 0] - C[:, 0], I[:, 1] - C[:, 1]
            d2 = dx*dx + dy*dy
            term = LUMINANCE*I[:, 2]/torch.where(d2 > 1, d2, torch.ones_like(d2))
            acc = torch.where(~blocked & self.imask[env, i], acc + term, acc)
        return torch.where(acc < 1, acc, torch.ones_like(acc))

    def render(self, agents):
        N, A, M, R, AF = self.N, self.A, self.M, self.R, self.A*self.M
        s, c = _sincospi(agents['angles']/180.)                                   # (N, A)
        p = agents['positions']
        # draw (kernels.cu:297-318)
        mx, my = self.model[None, None, :, :, 0], self.model[None, None, :, :, 1]  # (1, 1, M, 2)
        S, C_, P = s[..., None, None], c[..., None, None], p[:, :, None, None]
        drawn = torch.stack([C_*mx - S*my + P[..., 0], S*mx + C_*my + P[..., 1]], -1)   # (N, A, M, 2, 2)
        self.lines[:, :AF] = drawn.reshape(N, AF, 4)
        # rays (kernels.cu:234-236,334-344)
        rr = torch.arange(R, dtype=torch.float32)
        uy = (F(R) - 2*rr - 1)*self.half_screen/F(R)
        ru = torch.stack([c[..., None]*1. - s[..., None]*uy, s[..., None]*1. + c[..., None]*uy], -1)   # (N, A, R, 2)
        rlen = _len(ru)
        near = self.radius/rlen
        best_s = torch.full((N, A, R), float('inf'))
        best_i = torch.full((N, A, R), -1, dtype=torch.int64)
        best_loc = torch.full((N, A, R), float('nan'))
        best_dot = torch.full((N, A, R), float('nan'))
        P3 = p[:, :, None]
        for l in range(self.lines.shape[1]):                                      # the fold, in line order
            la, lb = self.lines[:, None, None, l, :2], self.lines[:, None, None, l, 2:]
            v = lb - la
            qs, qt = _intersect(P3, ru, la, v)
            dt = _dot(ru, v)/(rlen*_len(v) + 1.e-6)
            take = (0 <= qt) & (qt <= 1) & (near < qs) & (qs < best_s - 1.e-4) & self.lmask[:, l, None, None]
            best_s = torch.where(take, qs, best_s)
            best_i = torch.where(take, torch.full_like(best_i, l), best_i)
            best_loc = torch.where(take, qt, best_loc)
            best_dot = torch.where(take, dt, best_dot)
        # shade (kernels.cu:387-450)
        hit = best_i >= 0
        start = self.lines_starts[:, None, None] + best_i.clamp(min=0)
        w = self.tex_widths[start]
        wf = w.float()
        loc = torch.where(hit, best_loc, torch.zeros_like(best_loc))
        y = torch.minimum(loc*(wf + 1), wf - 1)
        fl = torch.where(y - 1 > 0, y - 1, torch.zeros_like(y)).to(torch.int64)
        fr = torch.minimum(y, wf - 1).to(torch.int64)
        ld, rd = (y - (fl + 1).float()).abs() + 1.e-3, (y - (fr + 1).float()).abs() + 1.e-3
        lw_, rw_ = rd/(ld + rd), ld/(ld + rd)
        ts = self.tex_starts[start]
        il, ir = (ts + fl).clamp(0, len(self.baked) - 1), (ts + fr).clamp(0, len(self.baked) - 1)
        intensity = lw_*self.baked[il] + rw_*self.baked[ir]
        dyn = hit & (best_i < AF)
        if dyn.any():
            env = dyn.nonzero()[:, 0]
            line = self.lines[env, best_i[dyn]]
            t = best_loc[dyn][:, None]
            Cpt = line[:, :2]*(1 - t) + line[:, 2:]*t
            intensity[dyn] = self._light_intensity(Cpt, env)
        dn = 1 - best_dot*best_dot
        screen = (dn*intensity)[..., None]*(lw_[..., None]*self.tex[il] + rw_[..., None]*self.tex[ir])
        screen = torch.where(hit[..., None], screen, torch.zeros_like(screen))
        return dict(indices=best_i.to(torch.int32), locations=best_loc, dots=best_dot, distances=best_s*rlen, screen=screen)


def step(world, agents):
    """One env step: physics then render."""
    progress = world.physics(agents)
    return progress, world.render(agents)
