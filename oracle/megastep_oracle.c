/*
 * megastep_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * A plain-C restatement of the reference's simulation hot path, i.e. of
 *   /root/reference/megastep/src/kernels.cu   (physics / bake / render)
 *   /root/reference/megastep/src/common.h     (Ragged index arrays)
 * evaluated in IEEE-754 binary32 exactly as the source is written (no FMA
 * contraction, correctly rounded / and sqrt, denormals kept).  Every function
 * cites the reference lines it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (megastep_amd/) never imports, links or
 * executes anything under oracle/.
 *
 * PARITY PINNING STATUS
 *   physics : pinned by the reference's one documented known answer
 *             (docs/tutorials/minimal-env/index.rst:140-145 -> x = 5.8649).
 *   ragged  : pinned by ragged.py:77-103 and docs/concepts.rst:205-219.
 *   render  : hit indices and hues pinned by the one rendered output the
 *             reference publishes, docs/tutorials/minimal-env/render.png (64
 *             columns: which wall each ray lands on, each wall's colour
 *             direction through texture, shading and gamma);
 *             tests/golden/make_docs_images.py, tests/test_oracle.py.
 *   render (distances, locations, dots, brightness) / bake : PARITY UNPINNED
 *             by the reference (it has no test, fixture or golden vector for
 *             them - the image's textures and light were drawn from an unseeded
 *             RNG - and its CUDA extension can be neither built nor run in the
 *             authoring container).  Pinned only by analytic closed forms
 *             (tests/test_oracle.py).
 *
 * Deliberate, documented definitions where the CUDA source leaves the bits to
 * the toolchain (all within the 1e-5 tolerance of BASELINE.json):
 *   - fminf/fmaxf     -> or_min/or_max below (NaN-ignoring, first operand on ties)
 *   - cospif/sinpif   -> or_sincospi: exact range reduction + double Taylor, rounded once
 *   - ATen `%`        -> fmod-based remainder (aten/src/ATen/native/cpu/BinaryOpsKernel.cpp)
 *   - --use_fast_math -> not modelled; IEEE arithmetic throughout
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define OR_OK 0
#define OR_EINVAL -1

/* ---- plain-pointer views of the reference's value types (common.h:157-226) ---- */
typedef struct {
    int n_envs, n_agents, n_model;       /* N, A, M (model.size(0))            */
    const float* lights_vals;            /* (sum I, 3) = x, y, intensity       */
    const int*   lights_widths;          /* (N,)                               */
    const int*   lights_starts;          /* (N,)                               */
    float*       lines_vals;             /* (sum L, 2, 2); agent rows rewritten by render */
    const int*   lines_widths;           /* (N,)                               */
    const int*   lines_starts;           /* (N,)                               */
    const int*   lines_inverse;          /* (sum L,) line -> env               */
    const float* textures_vals;          /* (sum T, 3)                         */
    const int*   textures_widths;        /* (sum L,) texels per line           */
    const int*   textures_starts;        /* (sum L,)                           */
    const int*   textures_inverse;       /* (sum T,) texel -> line             */
    const float* model;                  /* (M, 2, 2)                          */
    float*       baked_vals;             /* (sum T,)                           */
    int n_lines_total, n_lights_total, n_texels_total;
} OrScenery;

typedef struct {
    float* angles;       /* (N, A)    degrees   */
    float* positions;    /* (N, A, 2) metres    */
    float* angvelocity;  /* (N, A)    degrees/s */
    float* velocity;     /* (N, A, 2) metres/s  */
} OrAgents;

typedef struct {
    int*   indices;      /* (N, A, R)    */
    float* locations;    /* (N, A, R)    */
    float* dots;         /* (N, A, R)    */
    float* distances;    /* (N, A, R)    */
    float* screen;       /* (N, A, R, 3) */
} OrRender;

/* kernels.cu:12-27 `initialize` */
typedef struct {
    float agent_radius;
    int   res;
    float fov;
    float fps;
} OrConfig;

static const float OR_AMBIENT = .1f;     /* kernels.cu:9   */
static const float OR_LUMINANCE = 2.f;   /* kernels.cu:240 */

typedef struct { float x, y; } Pt;

static inline Pt pt(float x, float y) { Pt p = {x, y}; return p; }
static inline Pt pt_sub(Pt a, Pt b) { return pt(a.x - b.x, a.y - b.y); }          /* kernels.cu:46 */
static inline Pt pt_add(Pt a, Pt b) { return pt(a.x + b.x, a.y + b.y); }          /* kernels.cu:45 */
static inline Pt pt_mul(Pt a, float v) { return pt(a.x*v, a.y*v); }               /* kernels.cu:44 */
static inline Pt pt_div(Pt a, float v) { return pt(a.x/v, a.y/v); }               /* kernels.cu:43 */
static inline float pt_len2(Pt a) { return a.x*a.x + a.y*a.y; }                   /* kernels.cu:48 */
static inline float pt_len(Pt a) { return sqrtf(pt_len2(a)); }                    /* kernels.cu:49 */
static inline float cross(Pt v, Pt w) { return v.x*w.y - v.y*w.x; }               /* kernels.cu:59-61 */
static inline float dot(Pt v, Pt w) { return v.x*w.x + v.y*w.y; }                 /* kernels.cu:63-65 */

/* fminf / fmaxf with the zero-sign and NaN behaviour pinned down */
static inline float or_min(float a, float b) { if (a != a) return b; return (b < a) ? b : a; }
static inline float or_max(float a, float b) { if (a != a) return b; return (b > a) ? b : a; }

/* kernels.cu:67-89 `intersect` */
typedef struct { float s, t; } Isect;
static inline Isect intersect(Pt P, Pt U, Pt Q, Pt V) {
    Isect r;
    const float UxV = cross(U, V);
    if (fabsf(UxV) < 1.e-3f) {
        r.s = INFINITY; r.t = INFINITY;
    } else {
        const Pt PQ = pt_sub(Q, P);
        r.s = cross(PQ, V)/UxV;
        r.t = cross(PQ, U)/UxV;
    }
    return r;
}

/* kernels.cu:91-107 `project` */
typedef struct { float s, d; } Proj;
static inline Proj project(Pt P, Pt U, Pt Q) {
    Proj r;
    const float u = pt_len(U) + 1e-6f;
    const Pt PQ = pt_sub(Q, P);
    r.s = dot(PQ, U)/(u*u);
    r.d = fabsf(cross(PQ, U))/u;
    return r;
}

/* kernels.cu:109-118 `sensibilize`: NaN -> 0, clamp(0.99 p, 0, 1). Written so the
 * result is never -0 (the CUDA fmaxf(-0, 0) sign is implementation-defined). */
static inline float sensibilize(float p) {
    const float q = p*.99f;
    if (!(q > 0.f)) return 0.f;   /* NaN, <= 0 and -0 all land here */
    return (q < 1.f) ? q : 1.f;
}

/* kernels.cu:119-133 circle-circle `collision` */
static inline float collision_cc(Pt p0, Pt v0, Pt p1, Pt v1, float agent_radius) {
    const float r = 1.001f*2.f*agent_radius;
    float x = 1.f;
    const Pt dv = pt_sub(v0, v1);
    const Proj a = project(p0, dv, p1);
    if ((0 < a.s) & (a.d < r)) {
        const float backoff = sqrtf(r*r - a.d*a.d)/pt_len(dv);
        x = or_min(x, sensibilize(a.s - backoff));
    }
    return x;
}

/* kernels.cu:135-171 circle-segment `collision` */
static inline float collision_cs(Pt p, Pt v, Pt la, Pt lb, float agent_radius) {
    const float r = 1.001f*agent_radius;
    float x = 1.f;
    const Pt lv = pt_sub(lb, la);

    /* passing through l (:143-146) */
    const Isect mid = intersect(p, v, la, lv);
    if ((0 < mid.s) & (mid.s < 1) & (0 < mid.t) & (mid.t < 1)) {
        x = or_min(x, sensibilize((1 - r/project(la, lv, p).d)*mid.s));
    }
    /* within r of l.a (:149-153) */
    const Proj a = project(p, v, la);
    if ((0 < a.s) & (a.d < r)) {
        const float backoff = sqrtf(r*r - a.d*a.d)/pt_len(v);
        x = or_min(x, sensibilize(a.s - backoff));
    }
    /* within r of l.b (:156-160) */
    const Proj b = project(p, v, lb);
    if ((0 < b.s) & (b.d < r)) {
        const float backoff = sqrtf(r*r - b.d*b.d)/pt_len(v);
        x = or_min(x, sensibilize(b.s - backoff));
    }
    /* within r of the middle of l (:163-168) */
    const Proj side = project(la, lv, pt_add(p, v));
    if ((0 < side.s) & (side.s < 1) & (side.d < r)) {
        const float dp = project(la, lv, p).d;
        const float dq = side.d;
        x = or_min(x, sensibilize((dp - r)/(dp - dq)));
    }
    return x;
}

/* ATen `%` on float tensors == remainder(): fmod then sign fix-up */
static inline float or_remainder(float a, float b) {
    float m = fmodf(a, b);
    if ((m != 0.f) && ((b < 0.f) != (m < 0.f))) m += b;
    return m;
}
/* kernels.cu:173-175 */
static inline float normalize_degrees(float a) {
    return or_remainder(or_remainder(a, 360.f) + 180.f, 360.f) - 180.f;
}

/* sin(pi x), cos(pi x): stands in for CUDA's sinpif/cospif (kernels.cu:305-306,336-337).
 * Range reduction is exact in binary32; the kernel is a double Taylor series
 * (no FMA), rounded to binary32 once. */
void oracle_sincospi(float x, float* sp, float* cp) {
    /* exact in binary32: y = x - 2 rint(x/2) in [-1, 1], then z = y - rint(2y)/2 in [-1/4, 1/4] */
    const float y = x - 2.f*rintf(x*0.5f);
    const float nq = rintf(2.f*y);     /* nearest quarter-turn, ties to even */
    const float z = y - 0.5f*nq;
    const int q = ((int)nq) & 3;
    const double zd = (double)z;
    const double w = zd*zd;
    /* pi^(2k+1)/(2k+1)! and pi^(2k)/(2k)! */
    const double S0 = 3.141592653589793, S1 = -5.16771278004997, S2 = 2.5501640398773455,
                 S3 = -0.5992645293207921, S4 = 0.08214588661112823, S5 = -0.0073704309457143504,
                 S6 = 0.00046630280576761255, S7 = -2.1915353447830217e-05;
    const double C1 = -4.934802200544679, C2 = 4.0587121264167685, C3 = -1.3352627688545895,
                 C4 = 0.2353306303588932, C5 = -0.02580689139001406, C6 = 0.0019295743094039231,
                 C7 = -0.0001046381049248457, C8 = 4.303069587032947e-06;
    double ps = S7;
    ps = ps*w + S6; ps = ps*w + S5; ps = ps*w + S4; ps = ps*w + S3;
    ps = ps*w + S2; ps = ps*w + S1; ps = ps*w + S0;
    ps = ps*zd;
    double pc = C8;
    pc = pc*w + C7; pc = pc*w + C6; pc = pc*w + C5; pc = pc*w + C4;
    pc = pc*w + C3; pc = pc*w + C2; pc = pc*w + C1;
    pc = pc*w + 1.0;
    const float S = (float)ps, C = (float)pc;
    float s, c;
    switch (q) {
        case 0:  s =  S; c =  C; break;
        case 1:  s =  C; c = -S; break;
        case 2:  s = -S; c = -C; break;
        default: s = -C; c =  S; break;
    }
    *sp = s; *cp = c;
}

/* common.h:112-128 Ragged ctor + common.h:91-98 `inverses` */
int oracle_ragged_index(const int* widths, int W, int* starts, int* ends, int* inverse) {
    int acc = 0;
    for (int i = 0; i < W; i++) {
        if (widths[i] < 0) return OR_EINVAL;
        starts[i] = acc;
        for (int j = 0; j < widths[i]; j++) inverse[acc + j] = i;
        acc += widths[i];
        ends[i] = acc;
    }
    return OR_OK;
}

/* ------------------------------------------------------------------ physics */
/* kernels.cu:179-230: collision_kernel + the ATen epilogue */
int oracle_physics(const OrScenery* sc, OrAgents* ag, float* progress, const OrConfig* cfg) {
    const int N = sc->n_envs, A = sc->n_agents, DF = sc->n_agents*sc->n_model;
    const float fps = cfg->fps, R_ = cfg->agent_radius;
    #pragma omp parallel for schedule(dynamic, 1)
    for (int n = 0; n < N; n++) {
        const int L = sc->lines_widths[n];
        const float* lines = sc->lines_vals + 4*(size_t)sc->lines_starts[n];
        for (int d0 = 0; d0 < A; d0++) {
            const Pt p0 = pt(ag->positions[(n*A + d0)*2], ag->positions[(n*A + d0)*2 + 1]);
            const Pt m0 = pt(ag->velocity[(n*A + d0)*2], ag->velocity[(n*A + d0)*2 + 1]);
            float x = 1.f;
            for (int d1 = 0; d1 < A; d1++) {
                if (d0 != d1) {
                    const Pt p1 = pt(ag->positions[(n*A + d1)*2], ag->positions[(n*A + d1)*2 + 1]);
                    const Pt m1 = pt(ag->velocity[(n*A + d1)*2], ag->velocity[(n*A + d1)*2 + 1]);
                    x = or_min(x, collision_cc(p0, pt_div(m0, fps), p1, pt_div(m1, fps), R_));
                }
            }
            for (int l = DF; l < L; l++) {
                const Pt la = pt(lines[4*l], lines[4*l + 1]), lb = pt(lines[4*l + 2], lines[4*l + 3]);
                x = or_min(x, collision_cs(p0, pt_div(m0, fps), la, lb, R_));
            }
            progress[n*A + d0] = x;
        }
    }
    /* epilogue, kernels.cu:224-227 (reads all of progress, so a second pass) */
    #pragma omp parallel for
    for (int i = 0; i < N*A; i++) {
        const float x = progress[i];
        ag->positions[2*i]     = ag->positions[2*i]     + x*ag->velocity[2*i]/fps;
        ag->positions[2*i + 1] = ag->positions[2*i + 1] + x*ag->velocity[2*i + 1]/fps;
        if (x < 1) { ag->velocity[2*i] = 0.f; ag->velocity[2*i + 1] = 0.f; }
        ag->angles[i] = normalize_degrees(ag->angles[i] + x*ag->angvelocity[i]/fps);
        if (x < 1) ag->angvelocity[i] = 0.f;
    }
    return OR_OK;
}

/* ------------------------------------------------------------------ lighting */
/* kernels.cu:238-268 `light_intensity` */
static float light_intensity(const OrScenery* sc, Pt C, int n, int af) {
    float intensity = OR_AMBIENT;
    const int num_i = sc->lights_widths[n], num_l = sc->lines_widths[n];
    const float* lights = sc->lights_vals + 3*(size_t)sc->lights_starts[n];
    const float* lines = sc->lines_vals + 4*(size_t)sc->lines_starts[n];
    for (int i = 0; i < num_i; i++) {
        const Pt I = pt(lights[3*i], lights[3*i + 1]);
        const float Ii = lights[3*i + 2];
        int unobstructed = 1;
        for (int l1 = af; l1 < num_l; l1++) {
            const Pt la = pt(lines[4*l1], lines[4*l1 + 1]), lb = pt(lines[4*l1 + 2], lines[4*l1 + 3]);
            const Isect p = intersect(I, pt_sub(C, I), la, pt_sub(lb, la));
            const int obstructed = (p.t > 0.f) & (p.t < 1.f) & (p.s > 0.f) & (p.s < .999f);
            unobstructed = unobstructed & !obstructed;
        }
        const float d2 = pt_len2(pt_sub(I, C));
        if (unobstructed) intensity += OR_LUMINANCE*Ii/or_max(d2, 1.f);
    }
    return or_min(intensity, 1.f);
}

/* kernels.cu:270-293 `baking_kernel` / `bake` */
int oracle_bake(OrScenery* sc, const OrConfig* cfg) {
    (void)cfg;
    const int T = sc->n_texels_total, af = sc->n_agents*sc->n_model;
    #pragma omp parallel for schedule(dynamic, 256)
    for (int t = 0; t < T; t++) {
        const int l0 = sc->textures_inverse[t];
        const int n = sc->lines_inverse[l0];
        const float loc = ((unsigned)(t - sc->textures_starts[l0]) + .5f)/sc->textures_widths[l0];
        const float* ln = sc->lines_vals + 4*(size_t)l0;
        const Pt C = pt_add(pt_mul(pt(ln[0], ln[1]), 1.f - loc), pt_mul(pt(ln[2], ln[3]), loc));
        sc->baked_vals[t] = light_intensity(sc, C, n, af);
    }
    return OR_OK;
}

/* ------------------------------------------------------------------ render */
/* kernels.cu:234-236 */
static inline float ray_y(float r, float R, float half_screen) { return (R - 2*r - 1)*half_screen/R; }

/* kernels.cu:387-405 `filter` */
typedef struct { int l, r; float lw, rw; } Filt;
static inline Filt filter(float x, int w) {
    Filt f;
    const float y = or_min(x*(w + 1), (float)(w - 1));
    f.l = (int)or_max(y - 1, 0.f);
    f.r = (int)or_min(y, (float)(w - 1));
    const float ld = fabsf(y - (f.l + 1)) + 1.e-3f;
    const float rd = fabsf(y - (f.r + 1)) + 1.e-3f;
    f.lw = rd/(ld + rd);
    f.rw = ld/(ld + rd);
    return f;
}

/* kernels.cu:452-475 `render` = draw_kernel (:297-318) -> raycast_kernel (:326-383) -> shader_kernel (:407-450) */
int oracle_render(OrScenery* sc, const OrAgents* ag, OrRender* out, const OrConfig* cfg) {
    const int N = sc->n_envs, A = sc->n_agents, M = sc->n_model, R = cfg->res, AF = A*M;
    /* kernels.cu:22 */
    const float half_screen = tanf(3.14159265358979323846f/180.f*cfg->fov/2.);
    const float R_ = cfg->agent_radius;

    #pragma omp parallel for schedule(dynamic, 1)
    for (int n = 0; n < N; n++) {
        float* lines = sc->lines_vals + 4*(size_t)sc->lines_starts[n];
        const int num_l = sc->lines_widths[n];

        /* draw_kernel */
        for (int a = 0; a < A; a++) {
            float s, c;
            oracle_sincospi(ag->angles[n*A + a]/180.f, &s, &c);
            const float px = ag->positions[(n*A + a)*2], py = ag->positions[(n*A + a)*2 + 1];
            for (int m = 0; m < M; m++) for (int e = 0; e < 2; e++) {
                const float mx = sc->model[(m*2 + e)*2], my = sc->model[(m*2 + e)*2 + 1];
                lines[((a*M + m)*2 + e)*2]     = c*mx - s*my + px;
                lines[((a*M + m)*2 + e)*2 + 1] = s*mx + c*my + py;
            }
        }

        for (int a = 0; a < A; a++) {
            float s, c;
            oracle_sincospi(ag->angles[n*A + a]/180.f, &s, &c);
            const Pt p = pt(ag->positions[(n*A + a)*2], ag->positions[(n*A + a)*2 + 1]);
            for (int r = 0; r < R; r++) {
                /* raycast_kernel */
                const Pt u = pt(1.f, ray_y((float)r, (float)R, half_screen));
                const Pt ru = pt(c*u.x - s*u.y, s*u.x + c*u.y);
                const float rlen = pt_len(ru);
                float nearest_idx = -1;
                float nearest_s = INFINITY, nearest_loc = NAN, nearest_dot = NAN;
                for (int l = 0; l < num_l; l++) {
                    const Pt la = pt(lines[4*l], lines[4*l + 1]), lb = pt(lines[4*l + 2], lines[4*l + 3]);
                    const Pt v = pt(lb.x - la.x, lb.y - la.y);
                    const Isect q = intersect(p, ru, la, pt_sub(lb, la));
                    const float dtop = dot(ru, v);
                    const float dbot = rlen*pt_len(v);
                    const float dt = dtop/(dbot + 1.e-6f);
                    const int hit = (0 <= q.t) & (q.t <= 1);
                    const int better = (R_/rlen < q.s) & (q.s < nearest_s - 1.e-4f);
                    if (hit & better) {
                        nearest_s = q.s;
                        nearest_idx = (float)l;
                        nearest_loc = q.t;
                        nearest_dot = dt;
                    }
                }
                const size_t o = ((size_t)n*A + a)*R + r;
                out->indices[o] = (int)nearest_idx;
                out->locations[o] = nearest_loc;
                out->dots[o] = nearest_dot;
                out->distances[o] = nearest_s*rlen;

                /* shader_kernel */
                float s0 = 0.f, s1 = 0.f, s2 = 0.f;
                const int l0 = out->indices[o];
                if (l0 >= 0) {
                    const float loc = nearest_loc;
                    const int start = sc->lines_starts[n] + l0;
                    const Filt f = filter(loc, sc->textures_widths[start]);
                    const float* tex = sc->textures_vals + 3*(size_t)sc->textures_starts[start];
                    const float* bk = sc->baked_vals + (size_t)sc->textures_starts[start];
                    const float* tl = tex + 3*f.l; const float* tr = tex + 3*f.r;
                    float intensity;
                    if (l0 < AF) {
                        const Pt C = pt_add(pt_mul(pt(lines[4*l0], lines[4*l0 + 1]), 1 - loc),
                                            pt_mul(pt(lines[4*l0 + 2], lines[4*l0 + 3]), loc));
                        intensity = light_intensity(sc, C, n, AF);
                    } else {
                        intensity = f.lw*bk[f.l] + f.rw*bk[f.r];
                    }
                    const float dn = 1 - nearest_dot*nearest_dot;
                    s0 = dn*intensity*(f.lw*tl[0] + f.rw*tr[0]);
                    s1 = dn*intensity*(f.lw*tl[1] + f.rw*tr[1]);
                    s2 = dn*intensity*(f.lw*tl[2] + f.rw*tr[2]);
                }
                out->screen[3*o] = s0; out->screen[3*o + 1] = s1; out->screen[3*o + 2] = s2;
            }
        }
    }
    return OR_OK;
}

/* exposed for the unit tests of the scalar helpers */
float oracle_collision_cs(float px, float py, float vx, float vy, float ax, float ay, float bx, float by, float radius) {
    return collision_cs(pt(px, py), pt(vx, vy), pt(ax, ay), pt(bx, by), radius);
}
float oracle_collision_cc(float p0x, float p0y, float v0x, float v0y, float p1x, float p1y, float v1x, float v1y, float radius) {
    return collision_cc(pt(p0x, p0y), pt(v0x, v0y), pt(p1x, p1y), pt(v1x, v1y), radius);
}
float oracle_normalize_degrees(float a) { return normalize_degrees(a); }
void oracle_filter(float x, int w, int* l, int* r, float* lw, float* rw) {
    const Filt f = filter(x, w); *l = f.l; *r = f.r; *lw = f.lw; *rw = f.rw;
}
int oracle_abi_version(void) { return 1; }
