// kernels/render.h -- render_prep_kernel, render_kernel<IMPL, RW, OBS, SHADE, NG, STEP>, dynlight_kernel.
// Part of megastep_hip.hip's one translation unit (included there, inside its anonymous namespace, in this order: math,
// physics, lighting, render, bake, wallgrid); not a header to compile on its own.
// ------------------------------------------------------------------------------------------------
// render = draw + raycast + shader                                            kernels.cu:297-475
// ------------------------------------------------------------------------------------------------
#ifndef MS_GROUPS
#define MS_GROUPS 8
#endif
// A/B knobs of the pair raycasts (tools/ab_variants.sh builds one library per setting; the defaults are the product):
//   MS_V1_OPTS  bit 0: IMPL 1 takes IMPL 2's interval arithmetic (no clipping); bit 1: IMPL 1 takes IMPL 2's single
//               atomic + hysteresis flag instead of the three-slot cascade
//   MS_V2_OPTS  bit 0: IMPL 2 drains its list after every chunk; bit 1: IMPL 2 clips like IMPL 1
//               (tried and dropped: keys from v_rcp_f32 with the exact quotient once per ray - correct, not faster)
//   MS_AB_IMPLS 1: the library also holds the two older raycasts (IMPL 1 "pairs": per-chunk pair windows; IMPL 0 "seq":
//               the reference's fold in its literal order, every line, no lists), selected per call by the environment
//               variable MEGASTEP_RENDER_IMPL=pairs|seq - `make ab` builds it as libmegastep_hip_ab.so; the product
//               library holds IMPL 2 alone and reads no environment on its hot path
#ifndef MS_AB_IMPLS
#define MS_AB_IMPLS 0
#endif
#ifndef MS_V1_OPTS
#define MS_V1_OPTS 0
#endif
#ifndef MS_V2_OPTS
#define MS_V2_OPTS 0
#endif


constexpr int GROUPS = MS_GROUPS;     // ray groups (sub-wedges) per wave
constexpr int GSIZE = WAVE/GROUPS;    // rays per group
constexpr int PAIRS = 128;            // capacity of a wave's (wall, light) pair list in the dynamic-light pass

struct Cand { float pqx, pqy, vx, vy; };     // ray-independent half of intersect(), read as one b128

// The drawn (world-frame) model line `l` of env n: draw_kernel, kernels.cu:297-318.
// sin/cos of a heading where it is not worth a copy of the code: (sin(pi x), cos(pi x)), as sincospi_f gives them
__device__ __attribute__((noinline)) float2 sincospi_called(const float x) {
    float s_, c_;
    sincospi_f(x, s_, c_);
    return make_float2(s_, c_);
}

// (The work is in a function that is not inlined and takes plain pointers: it serves sceneries with more than 64 agents
// per env only, and a copy of its binary64 sin/cos at each of the render kernel's half-dozen call sites is code every
// wave would have to be fetched past.)
__device__ __attribute__((noinline)) float4 drawn_line_of(const float* angles, const float* positions, const float* model,
                                                          const int n_agents, const int M, const int n, const int l) {
    const int a = l / M, m = l - a*M;
    float s, c;
    sincospi_f(angles[n*n_agents + a]/180.f, s, c);
    const float2 p = reinterpret_cast<const float2*>(positions)[n*n_agents + a];
    const float4 mdl = reinterpret_cast<const float4*>(model)[m];
    float4 w;
    w.x = c*mdl.x - s*mdl.y + p.x;
    w.y = s*mdl.x + c*mdl.y + p.y;
    w.z = c*mdl.z - s*mdl.w + p.x;
    w.w = s*mdl.z + c*mdl.w + p.y;
    return w;
}
__device__ inline float4 drawn_line(const MsScenery& sc, const MsAgents& ag, int n, int l) {
    return drawn_line_of(ag.angles, ag.positions, sc.model, sc.n_agents, sc.n_model, n, l);
}

// kernels.cu:394-405
struct Filt { int l, r; float lw, rw; };
__device__ inline Filt tex_filter(float x, int w) {
    Filt f;
    const float y = ms_min(x*(w + 1), (float)(w - 1));
    f.l = (int)ms_max(y - 1, 0.f);
    f.r = (int)ms_min(y, (float)(w - 1));
    const float ld = fabsf(y - (f.l + 1)) + 1.e-3f;
    const float rd = fabsf(y - (f.r + 1)) + 1.e-3f;
#if MS_DIV_INRANGE
    const float den = ld + rd, r_den = rcp_refined(den);                // (in [2e-3, 2 w]: see div_inrange; one reciprocal for both)
    f.lw = div_by_refined(rd, den, r_den);
    f.rw = div_by_refined(ld, den, r_den);
#else
    f.lw = rd/(ld + rd);
    f.rw = ld/(ld + rd);
#endif
    return f;
}

// First launch of ms_render when a workspace is given: zeroes the queue counter and evaluates every agent's
// sin/cos (binary64 inside, see sincospi_f) once, instead of once per wavefront of the raycast.
// Workspace layout: [0] queue length, [1] rays that took the sequential fold, [2] wavefronts that took its lane-parallel
// form (telemetry for tests) | [16, 16 + N A ceil(R/64)) queued ray groups | (8-byte aligned) (sin, cos) per (env, agent), at
// `headings_at` = 16 + that count rounded up to even - a place that depends on the shapes alone, as MS_RENDER_WORKSPACE_INTS
// does, NOT on how many blocks the launch has: with several ray groups a wave every XCD gets as many blocks as the fullest
// one needs, which can be more than N A ceil(R/64) (round 4 put the headings behind the block count: past the end then).
// Launch-invariant values the host works out once per ms_render call instead of every wave doing so on the VALU:
// culling constants, and exact unsigned division by F = A*G, G and M via multiply-high (Granlund & Montgomery).
// sqrtf() for an argument known to be a normal number (not zero, denormal, infinite or NaN): the correctly rounded root
// the compiler's own expansion gives (v_sqrt_f32 is good to 1 ulp; the residuals of its two neighbours decide) without
// that expansion's rescaling of tiny arguments and its special cases - 8 instructions of 20.
#ifndef MS_SQRT_NORMAL
#define MS_SQRT_NORMAL 1               // (0: sqrtf() for the rays' lengths - the A/B: 34.5 -> 34.3 us at the headline, same bits)
#endif
__device__ inline float sqrt_normal(const float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float dn = bits_f(f_bits(s) - 1u), up = bits_f(f_bits(s) + 1u);
    const float r_dn = __builtin_fmaf(-dn, s, x), r_up = __builtin_fmaf(-up, s, x);
    float r = (r_dn <= 0.f) ? dn : s;
    r = (r_up > 0.f) ? up : r;
    return r;
}

// sqrtf() where the argument may be anything: sqrt_normal for normal numbers (and zero: see there), the library's for the rest
// (denormals, infinities, NaNs - a wall shorter than 10^-19 m) behind a branch no wave takes.
__device__ __attribute__((noinline)) float sqrtf_called(const float x) { return sqrtf(x); }
__device__ inline float sqrt_any(const float x) {
#if MS_DIV_INRANGE
    float r = sqrt_normal(x);
    if (__builtin_expect(!(x >= 1.e-30f) && x != 0.f, 0) || __builtin_expect(!(x <= 1.e30f), 0)) r = sqrtf_called(x);
    return r;
#else
    return sqrtf(x);
#endif
}

struct RenderConsts {
    float x_clip, c_b;
    Divisor by_f, by_g, by_m;
    Divisor by_f1, by_g1;          // render_kernel's NG > 1: the waves of one ray group at the end of every XCD's blocks (see there),
    int envs_lo, envs_rem, tail;   //   an XCD's envs (n_envs/8, the first n_envs % 8 XCDs one more) and how many of them those waves take
    int skip_own;                  // the agent's own model lines lie inside its near plane: no ray of its can hit them
    float inv_res;                 // 1/res where that is a power of two (x/res is then x*inv_res bit for bit), else 0
    int telemetry;                 // ms_debug_pair_telemetry: pair / window counts into workspace[3], [4]
    int ws_headings;               // where in MsRender.workspace the (sin, cos) pairs of render_prep_kernel start, in 4-byte words
};
// ... and what the STEP = 1 instantiations (ms_step_render: physics and render of a single-agent env as one wave's work) take on
// top: as a type of its own, so that the kernel-argument segment of every other instantiation is byte for byte what it was
// (they sit at 80 registers and 150 spilled scalars: four more bytes of arguments once cost the widest of them 5 %).
struct RenderConstsStep : RenderConsts {
    float* progress;               // (N, 1): MsPhysics' output
    float fps;
    const unsigned* wg_cells_physics;   // the wall grid's cell headers as ms_step_physics would be given them (ms_render's copy of the scenery
                                        // drops them when the call's near plane / field of view is outside what the vis lists were built for)
    MsMovement mv;                 // ms_step_physics' optional prologue and bookkeeping, as it takes them (all-NULL structs: none)
    MsStepExtras ex;
};
// Which of MsRender's optional outputs are there, as bits - for the COLOURLESS instantiations, which ask a scalar register the
// wave has had since its first instruction whether an output is wanted and only fetch a pointer from the kernel-argument segment
// when it is: asked of the pointers themselves every check was a scalar load and a wait of its own, per group of rays, for
// outputs nobody wanted (512 rays, pooled depth alone: 121.8 -> 114.6 us, against 113.9 for bare distances).  ms_render hands
// the bits over in the upper half of MsRender.obs_subsample of the copy it launches with - NOT as a field of RenderConsts, and
// not to the colour instantiations: a 4-byte field more in the kernel's arguments, used or not, cost the colour kernels of four
// ray groups a wave 5 % (512 rays: 144.9 -> 152.0 us, same box, same sources otherwise - they sit at 80 registers and 150
// spilled scalars, and anything that moves their allocation moves their time; profiles/r05_ab_out_mask.txt).
enum { OUT_INDICES = 1, OUT_LOCATIONS = 2, OUT_DOTS = 4, OUT_DISTANCES = 8, OUT_SCREEN = 16, OUT_RGB = 32, OUT_DEPTH = 64, OUT_CENTRE = 128,
       OUT_SEEN = 256 };
// (test hook, ms_test_arithmetic: the shortcuts above next to the operations they stand for)
__global__ __launch_bounds__(WG) void arithmetic_test_kernel(const float* __restrict__ n, const float* __restrict__ d, float* __restrict__ q_inrange,
                                                             float* __restrict__ q_ieee, const float* __restrict__ x, float* __restrict__ r_any,
                                                             float* __restrict__ r_ieee, const long long count) {
    const long long i = (long long)blockIdx.x*WG + threadIdx.x;
    if (i >= count) return;
    if (q_inrange) q_inrange[i] = div_by_refined(n[i], d[i], rcp_refined(d[i]));
    if (q_ieee) q_ieee[i] = n[i]/d[i];
    if (r_any) r_any[i] = sqrt_any(x[i]);
    if (r_ieee) r_ieee[i] = sqrtf(x[i]);
}

// Which rays of which agent the one-wave block `b` of a render launch of `n_blocks` casts: env n, agent a, rays r0 .. r0 + span - 1
// (those below R).  False: a spare block (see below).  The kernel's own mapping - and, through ms_host_render_block, what
// tests/test_launch_geometry.py walks over whole launches on the CPU.
//   ng == 1: block = (env, agent, run of 64 rays), dealt so that hardware block b - which lands on XCD b % 8 - gives each XCD a
//     contiguous run of them: the fans of one env (and its lines) stay behind one L2.  (XCDs 0 .. n_blocks % 8 - 1 get a block
//     more; no branch: a branch ends the stretch of code hipcc gathers the kernel-argument loads of to its top.)
//   ng > 1: XCD x takes blocks x, x + 8, ... and is given a contiguous run of ENVS - an eighth of them, the first N mod 8 XCDs one
//     more - its blocks in two parts: waves of ng groups for its envs but the last rc.tail, and behind them waves of ONE group
//     for those.  A wave of four groups lives four times as long, and a launch of a few rounds of those ends with the machine
//     draining for most of one such life; the short waves are what the slots that come free take up then (render_plan sizes the
//     second part: about half a round of the long ones' work).  Every XCD has as many blocks as the one with the most envs
//     needs: one with an env fewer lets its last blocks go.
__host__ __device__ inline bool render_block(const int b, const int n_blocks, const int A, const int R, const int ng, const RenderConsts& rc,
                                             int& n, int& a, int& r0, int& span, int& fan) {
    const int xcd = b & 7, ix = b >> 3;
    fan = b;                                                             // (ng == 1: the run's number, env-major, that dynlight_kernel's queue holds)
    if (ng == 1) {
        const int q8 = n_blocks >> 3, r8 = n_blocks & 7;
        fan = xcd*q8 + (xcd < r8 ? xcd : r8) + ix;
        const int G = (R + WAVE - 1)/WAVE, F = A*G;   // g: which run of 64 rays of the agent's this wave casts
        n = div_by(fan, rc.by_f); const int rem = fan - n*F; a = div_by(rem, rc.by_g); r0 = (rem - a*G)*WAVE; span = WAVE;
        return true;
    }
    const int NR = ng*WAVE;
    const int e_x = rc.envs_lo + (xcd < rc.envs_rem ? 1 : 0), first_x = xcd*rc.envs_lo + (xcd < rc.envs_rem ? xcd : rc.envs_rem);
    const int t_x = rc.tail < e_x ? rc.tail : e_x;
    const int Gw = (R + NR - 1)/NR, w_x = (e_x - t_x)*A*Gw;
    const bool single = ix >= w_x;
    const int f = single ? ix - w_x : ix;
    const int G = single ? (R + WAVE - 1)/WAVE : Gw, F = A*G;
    const Divisor df = Divisor{single ? rc.by_f1.mul : rc.by_f.mul, single ? rc.by_f1.sh1 : rc.by_f.sh1, single ? rc.by_f1.sh2 : rc.by_f.sh2};
    const Divisor dg = Divisor{single ? rc.by_g1.mul : rc.by_g.mul, single ? rc.by_g1.sh1 : rc.by_g.sh1, single ? rc.by_g1.sh2 : rc.by_g.sh2};
    const int nn = div_by(f, df), rem = f - nn*F;
    if (nn >= (single ? t_x : e_x - t_x)) return false;
    a = div_by(rem, dg);
    n = first_x + (single ? e_x - t_x : 0) + nn;
    span = single ? WAVE : NR;
    r0 = (rem - a*G)*span;
    return true;
}

__global__ __launch_bounds__(WG) void render_prep_kernel(const MsAgents ag, int* __restrict__ workspace,
                                                         const int n_agents_total, const int headings_at) {
    const int i = blockIdx.x*WG + threadIdx.x;
    if (i == 0) { workspace[0] = 0; workspace[1] = 0; workspace[2] = 0; workspace[3] = 0; workspace[4] = 0; }
    if (i < n_agents_total) {
        float s, c;
        sincospi_f(ag.angles[i]/180.f, s, c);
        reinterpret_cast<float2*>(workspace + headings_at)[i] = make_float2(s, c);
    }
}

// The render kernel's parameter list as a struct, and a pointer to the kernel-argument segment typed as one.  What the
// kernel only needs at its end - texture and baked-light pointers, the light grid, the output planes - is read through
// this pointer THERE: as plain parameters hipcc loads them at the top, runs out of scalar registers, and parks them in
// vector-register lanes, which costs two memory round trips (a parked value has to have arrived) and ~40 instructions
// per wave before the first ray is cast.  The asm statement keeps the loads from being hoisted back up.
// How the per-ray planes are written: as non-temporal stores - nobody in this launch reads them back, and at 512 rays
// they are 235 MB per launch that would otherwise push the lines and textures out of the L2 (512 rays: 231 -> 212 us,
// 16384 envs x 64 rays: 149.5 -> 145.5 us, no difference at the headline shape).  (-DMS_NT_STORES=0: A/B knob)
#ifndef MS_NT_STORES
#define MS_NT_STORES 1
#endif
#if MS_NT_STORES
#define MS_OUT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define MS_OUT_STORE(v, p) (*(p) = (v))
#endif

struct RenderArgs { MsScenery sc; MsAgents ag; MsRender out; float agent_radius, half_screen; int R, n_fans; RenderConsts rc; };
static_assert(offsetof(RenderArgs, ag) == sizeof(MsScenery) && offsetof(RenderArgs, n_fans) + 4 == offsetof(RenderArgs, rc),
              "RenderArgs must mirror render_kernel's parameters");
typedef const RenderArgs __attribute__((address_space(4)))* LateArgs;
__device__ inline LateArgs late_args() {
    LateArgs p = (LateArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// Conservative interval [lo, lo + len) of a wave's rays that can hit a line, from the agent-frame coordinates of its
// ends (x forward, y left; c_a - (y/x) c_b is the continuous ray index).  Everything here only feeds the cull, whose
// margin is 10^5 roundings wide: fused multiply-adds and approximate reciprocals are fine.
//   CLIP = 1: an end behind the near clip plane is clipped to it;
//   CLIP = 0: it is replaced by the edge of the fan on the side the line leaves by (the sign of cross(a, b)) - the same
//             interval unless the line crosses the clip plane within centimetres of the agent, for fewer instructions.
// A ray's nearest hit out of hits that arrive in any order (render_kernel, pass 2) - the reference folds them in LINE order
// with a hysteresis, `if (near < s && s < x - 1e-4) x = s` (kernels.cu:369-376), so its answer depends on that order.
// A hit is a key (s bits << 32 | line): s > 0, so keys order by s, ties by line.  Three slots per ray hold the least keys
// seen - the second and third only fed by losers within 4e-4 of what beat them, which is all that can matter to the
// hysteresis.  hit_resolve: with (m, j*) the least key and m2 the runner-up's s, if m < m2 - 1e-4 then when the fold
// reaches j* its state is inf or some s_k >= m2, so j* takes over, and nothing later can pass `s < m - 1e-4`: the fold
// ends on (m, j*).  Otherwise the two best sit inside the band (a ray through a shared corner, coincident walls): with
// the third-best clearly behind, the fold of those two in line order settles it; failing that the caller redoes the ray
// by the literal fold (returns true).
// These four are the merge and the resolution of render_kernel's pass 2 word for word - there they stay written out in
// place (as calls they changed the register allocation of the whole kernel, and it is tuned to the last register); here
// they serve ms_host_fold_hits, with which tests/test_wallgrid.py plays hits in random orders, lockstep window by
// window as a wave does, against the literal fold.
__host__ inline uint32_t host_f_bits(const float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
__host__ inline float host_bits_f(const uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
__host__ inline unsigned long long slot_min(unsigned long long* slot, const unsigned long long key) {   // atomicMin's stand-in
    const unsigned long long old = *slot;
    if (key < old) *slot = key;
    return old;
}
__host__ inline unsigned long long hit_key(const float sv, const int line) { return ((unsigned long long)host_f_bits(sv) << 32) | (unsigned)line; }
// after slot_min on the first slot returned `old`: was there a hit before this one, and does the loser of the merge go on
// to the second slot?
__host__ inline bool hit_loser_matters(const unsigned long long key, const float sv, const unsigned long long old, unsigned long long& lose1) {
    const unsigned oh = (unsigned)(old >> 32);
    if (oh == 0xffffffffu) return false;
    const bool won = key < old;
    const float so = host_bits_f(oh);
    const float front = won ? sv : so, back = won ? so : sv;
    lose1 = won ? old : key;
    return back < front + 4.e-4f;
}
__host__ inline bool hit_resolve(const unsigned long long best, const unsigned long long second, const unsigned long long third,
                                 float& nearest_s, int& nearest_idx) {
    bool ambiguous = false;
    if (best != ~0ull) {
        const float s1 = host_bits_f((uint32_t)(best >> 32)), s2 = host_bits_f((uint32_t)(second >> 32)), s3 = host_bits_f((uint32_t)(third >> 32));
        const int i1 = (int)(uint32_t)best, i2 = (int)(uint32_t)second;
        nearest_s = s1;
        nearest_idx = i1;
        if ((second != ~0ull) && !(s1 < s2 - 1.e-4f)) {
            if ((third == ~0ull) || (s2 < s3 - 1.e-4f)) {
                const bool first_is_1 = i1 < i2;
                const float sa = first_is_1 ? s1 : s2, sb = first_is_1 ? s2 : s1;
                const int ia = first_is_1 ? i1 : i2, ib = first_is_1 ? i2 : i1;
                const bool b_wins = sb < sa - 1.e-4f;
                nearest_s = b_wins ? sb : sa;
                nearest_idx = b_wins ? ib : ia;
            } else {
                ambiguous = true;
            }
        }
    }
    return ambiguous;
}

// (the hardware's approximate reciprocal on the device, a division on the host - whose instantiations of the culls exist
// for the CPU tests: everything that goes through here only feeds margins that are thousands of roundings wide)
__host__ __device__ inline float rcp_approx(const float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.f/x;
#endif
}
// agent-frame coordinates (x forward, y left) of a line's two ends, given relative to the agent: PQ = a - p, DB = b - p
__host__ __device__ inline void agent_frame(const float cs, const float sn, const float pqx, const float pqy, const float dbx, const float dby,
                                            float& xa, float& ya, float& xb, float& yb) {
    xa = __builtin_fmaf(cs, pqx, sn*pqy); ya = __builtin_fmaf(cs, pqy, -(sn*pqx));
    xb = __builtin_fmaf(cs, dbx, sn*dby); yb = __builtin_fmaf(cs, dby, -(sn*dbx));
}
template <int CLIP>
__host__ __device__ inline void ray_interval(float xa, float ya, float xb, float yb, const bool live, const float x_clip,
                                    const float c_a, const float c_b, const float g0, const float last_local, int& lo, int& len,
                                    const float n_rays = 64.f) {
    const bool fa = xa >= x_clip, fb = xb >= x_clip;
    float ra, rb, marg;
    bool inc;
    if constexpr (CLIP == 1) {
        inc = fa | fb | !(xa == xa) | !(xb == xb);                      // wholly behind the clip plane: never hit
        if (fa != fb) {                                                 // clip the hidden end to x' = x_clip
            const float t = (x_clip - xa)*rcp_approx(xb - xa);
            const float yc = __builtin_fmaf(t, yb - ya, ya);
            if (fa) { xb = x_clip; yb = yc; } else { xa = x_clip; ya = yc; }
        }
        const float ysa = ya*rcp_approx(xa), ysb = yb*rcp_approx(xb);
        ra = __builtin_fmaf(-ysa, c_b, c_a); rb = __builtin_fmaf(-ysb, c_b, c_a);
        marg = __builtin_fmaf(1e-4f, fabsf(ra) + fabsf(rb), 0.05f);
    } else {
        inc = fa | fb;                                                  // (a NaN coordinate: the reference never hits such a line)
        const float ia = fa ? ya*rcp_approx(xa) : 0.f, ib = fb ? yb*rcp_approx(xb) : 0.f;
        const float fra = __builtin_fmaf(-ia, c_b, c_a), frb = __builtin_fmaf(-ib, c_b, c_a);
        marg = __builtin_fmaf(1e-4f, fabsf(fra) + fabsf(frb), 0.05f);
        const float edge = (xa*yb - ya*xb > 0.f) ? -INFINITY : INFINITY;   // from a towards b the ray index falls / rises
        ra = fa ? fra : -edge; rb = fb ? frb : edge;
    }
    // fminf/fmaxf drop NaNs towards the wide side, so a doubtful line keeps the full range
    const float flo = fminf(fmaxf(fminf(ra, rb) - (marg + g0), 0.f), n_rays);
    const float fhi = fmaxf(fminf(fmaxf(ra, rb) + (marg - g0), last_local), -1.f);
    lo = (int)ceilf(flo);
    const int n_ = (int)floorf(fhi) - lo + 1;
    len = (live & inc) ? (n_ > 0 ? n_ : 0) : 0;
}

// IMPL 0 ("seq"): every ray walks its group's line mask in index order - the reference's fold verbatim.
// IMPL 1 ("pairs"): (line, ray) pairs flattened over all 64 lanes + LDS atomic argmin; rays whose
//          two best hits sit inside the 1e-4 hysteresis band get the sequential fold.  Same bits, ~2x faster.
// RW = waves per workgroup.  The waves never talk to each other, so RW = 1 lets every wave give its slot and
// LDS back the moment it is done instead of waiting for the slowest of four.
// OBS = 1: any of the five per-ray outputs may be NULL, and pooled observations are written on request (the plain
// instantiation stays within 80 VGPRs: six waves per SIMD)
// OBS = 2 (round 6, with SHADE = 1): NONE of the five per-ray planes is wanted - only what the OBS = 1 epilogue can write besides
// them: pooled RGB-D, crosshair ids, first-sight books.  It is what both demo envs ask for (`modules.render(core, observers=...,
// fields=())`), every step, and in the OBS = 1 instantiation it was a question asked of five pointers per group of rays - a scalar
// load and a wait each - plus the staging of `screen` through LDS behind a branch, in a kernel of 150 spilled scalars: here the
// plane stores, their pointers and the staging are not in the kernel at all.
// SHADE = 0 (with OBS = 1): the caller wants no colour - neither `screen` nor pooled RGB (modules.Depth reads distances
// only, reference modules.py:170-184; BASELINE config 2 is depth-only).  Pass 3 is then not in the kernel at all: no
// texel row, no texel and baked-light gathers, no filter, no dynamic lighting of rays that landed on an agent - and the
// winning line itself is only fetched (for `locations`, `dots` or the first-sight books) if one of those is asked for:
// a distances-only wave ends with the raycast, without a single dependent load behind it.
// NG = 64-ray groups a wave serves (1, 2 or 4; IMPL 2 with one wave per workgroup).  At 128 rays and more an agent's
// waves each repeated the agent-side half of the work - state, cell, vis list and its arc cull, the agents' lines,
// pass 1 on every line their wedges share - and at 512 rays that was most of a wave's instructions on a chip whose
// vector ALUs were 0.99 busy.  A wave of NG groups does it once for 64 NG consecutive rays: pass 1 turns a line into an
// interval of all of them, pass 2 deals the (line, ray) pairs to the lanes whichever group the ray is in, and only the
// per-ray ends of the kernel - ray set-up, resolution, shading, stores - run group after group.  (ms_render picks NG from
// the resolution: 1 up to 64 rays - the headline's instantiation is what it was -, 2 up to 128, 4 beyond.)
// STEP = 1 (ms_step_render; one agent per env, up to 64 rays: the agent IS one wave): the wave runs its env's physics step first
// - kernels.cu:179-230 for one agent: reach, the near list of its cell (or every wall), the exact test, the integration epilogue -
// and renders from the pose it ends on.  Nothing crosses waves (no other wave reads this agent), so the step is ONE launch: the
// physics launch it replaces is 6 of C2's 17 us, 4 of them what launching any kernel behind another costs.
template <int IMPL, int RW, int OBS, int SHADE = 1, int NG = 1, int STEP = 0>
// Occupancy knobs of the render kernel (A/B builds; the defaults are the product): waves per SIMD the register allocation
// is held to, chunks of rows in flight, capacity of a wave's list of visible lines (which sizes its LDS block)
#ifndef MS_WAVES
#define MS_WAVES 6
#endif
#ifndef MS_ABLATE
#define MS_ABLATE 0                          // (instruction-count / time experiments: 4 ends a wave at once, 1 stops it after its set-up, 2 after pass 1 with
#endif                                       //  pass 2 skipped, 3 after the raycast; the outputs are then garbage)
// Chunks of 64 rows a walk has in flight per batch.  Three until round 5 - from before the wall grid, when a wave met every line
// of its env, five or six chunks of them; with the vis lists a batch is the agents' lines and the 30-odd walls that pass the arc
// cull: ONE chunk more often than not, and the other two slots' index arithmetic and (unconditional) row gathers were wasted
// on it.  One chunk a batch (profiles/r05_ab_ahead.txt, us per render launch, same box): headline 33.8 -> 33.3, C3 51.0 -> 49.4,
// 512 rays 141.0 -> 139.4, C5's share 190.0 -> 185.7, C2 11.8 -> 11.8; and 68 registers instead of 77.  (With them a seventh wave
// a SIMD fits once the list is cut to 85 lines - 5112 B, four LDS granules; round 3's "seven waves" still had five granules and
// never ran seven - and loses: 34.2 at the headline, 13.9 for 11.8 at C2, where a list of 85 lines overflows into second
// drains and the all-lines fold; so does the shorter list alone at six waves.  Six waves, 128 lines stay.)
#ifndef MS_AHEAD
#define MS_AHEAD 1
#endif
#ifndef MS_VCAP
#define MS_VCAP 128
#endif
// ... and of a wave of several ray groups (NG > 1), whose block is 30 bytes a line + 3072.  LDS is handed out in granules of
// 1280 bytes on gfx950: 128 lines (6912 B) cost six granules - 21 waves a CU where the registers allow 24 - and so do 112;
// 110 (6372 B) fit five.  Measured (tools/ab_groups.py, four groups a wave): 512 rays 147.5 -> 138.4 us at 96 lines (80: 138.9;
// 112, 120: 148), C5's share 201.0 -> 193.4 (80: 198.9 - its 1000-wall plans overflow a short list more often).
#ifndef MS_VCAP_WIDE
#define MS_VCAP_WIDE 110
#endif
// (MS_WAVES_WIDE: the same for the instantiations of several ray groups a wave - an A/B knob: at 5 the colour ones need no scratch)
#ifndef MS_WAVES_WIDE
#define MS_WAVES_WIDE 6
#endif
__global__ __launch_bounds__(RW*WAVE) __attribute__((amdgpu_waves_per_eu(NG > 1 ? MS_WAVES_WIDE : MS_WAVES, NG > 1 ? MS_WAVES_WIDE : MS_WAVES))) void render_kernel(
        const MsScenery sc, const MsAgents ag, const MsRender out,
        const float agent_radius, const float half_screen, const int R, const int n_fans,
        const std::conditional_t<STEP == 1, RenderConstsStep, RenderConsts> rc) {
    static_assert(STEP == 0 || (IMPL == 2 && RW == 1 && NG == 1), "the fused step: the product raycast, one wave per workgroup, one group of rays");
    // Per-wave LDS, one raw block so that the lighting at the end can reuse what the raycast is done with:
    //      0 cand   (64 x 16 B)  the chunk's 64 lines               | lighting: (wall, light) pair list, 2 KiB
    //   1024 ray    (64 x 16 B)  per ray: rx, ry, near              |
    //   2048 best   (64 x 8 B)   per ray: least key                 | lighting: shadow words, 512 B
    //   2560 second, 3072 third                                     |
    //   3584 info   (64 x 4 B)   per line: (first pair << 6) | first ray
    //   3840 mark   (64 x 4 B)   pair window: which line starts here
    //   4096 screen (192 x 4 B)  RGB staging
    // IMPL 2 lays its block out differently (see there): 6144 B
    PROBE_INIT
    static_assert(NG == 1 || (IMPL == 2 && RW == 1 && (NG == 2 || NG == 4)), "several ray groups per wave: the product raycast, one wave per workgroup");
    [[maybe_unused]] constexpr int NR = WAVE*NG;  // rays per wave
    // (NG > 1: the list is shared by the wave's groups and must outlive their epilogues, whose scratch - the lighting's pair
    // list and shadow words, the RGB staging - therefore sits in the per-group region behind it, O_EPI, not on top of it)
    constexpr int VCAP = NG == 1 ? MS_VCAP : MS_VCAP_WIDE;
    constexpr int O_EPI = (IMPL == 2 && NG > 1) ? 24*VCAP + 256 : 0;
    constexpr int LDS_PER_WAVE = IMPL != 2 ? 4864 : NG == 1 ? 24*VCAP + 3072 : O_EPI + 2816 + 6*VCAP;
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[RW][LDS_PER_WAVE];

    // (with one wave per workgroup the wave index is spelled out as 0: hipcc cannot tell that threadIdx.x >> 6 is, and
    // would otherwise keep env, agent, line count and every address derived from them in vector registers)
    const int tid = threadIdx.x, wave = RW == 1 ? 0 : tid >> 6;
    int lane = RW == 1 ? tid : tid & 63;         // (not const: see LANE_AFRESH)
    // In a loop over a wave's ray groups hipcc hoists everything that depends on the lane alone - a dozen LDS addresses,
    // masks, offsets - out of the loop and holds it in registers through all of it: 16-28 spilled to scratch memory at the
    // 80 the kernel is held to.  Made opaque at the top of every iteration, the lane is worked with afresh each time.
#define LANE_AFRESH asm volatile("" : "+v"(lane))
    Cand* const s_cand_w = reinterpret_cast<Cand*>(&s_raw[wave][0]);
    float* const s_screen_w = reinterpret_cast<float*>(&s_raw[wave][IMPL == 2 ? O_EPI : 4096]);   // (IMPL 2: the raycast is over by then)

    // (With one wave per workgroup the grid is exactly the blocks render_block() deals - ms_render launches it so: the count
    // comes from the kernel's own arguments, not from the dispatch packet, and at one group per wave there is no early exit -
    // either of which is a round trip of its own before the loads below may even be asked for.)
    const int b = blockIdx.x;
    [[maybe_unused]] int fan = b;                                        // (what the probe and the ablation builds label a wave with)
    const int A = sc.n_agents, AF = sc.n_agents*sc.n_model;
    int n, a, r0, span;
    if constexpr (RW == 1) {
        if (!render_block(b, n_fans, A, R, NG, rc, n, a, r0, span, fan)) return;
    } else {                                                             // (the A/B builds' older raycasts: several waves a workgroup)
        const int nb = (int)gridDim.x;
        const int q8 = nb >> 3, r8 = nb & 7, xcd = b & 7, ix = b >> 3;
        fan = (xcd*q8 + min(xcd, r8) + ix)*RW + wave;
        if (fan >= n_fans) return;                                       // waves are independent: no workgroup barriers below
        const int G = (R + WAVE - 1)/WAVE, F = A*G;
        n = div_by(fan, rc.by_f); const int rem = fan - n*F; a = div_by(rem, rc.by_g); r0 = (rem - a*G)*WAVE; span = WAVE;
    }
    if constexpr (MS_ABLATE == 4) return;                                // (what does a launch of this many one-wave workgroups cost on its own?)
#ifdef MS_PARK
    // (-DMS_PARK=<shader clocks>, an experiment: every render wave sits out that long before it starts, as it would at the
    // barrier of a single-launch step whose first wave does the env's physics - what do parked waves cost a launch?)
    { const long long t0_ = clock64(); while (clock64() - t0_ < MS_PARK) __builtin_amdgcn_s_sleep(8); }
#endif
    const int r = r0 + lane;                       // (this lane's ray in the wave's first group)
    const int r_last = min(r0 + span - 1, R - 1);
    [[maybe_unused]] const int n_live = r_last - r0 + 1;

    const int L = sc.lines_widths[n];
    const int base = sc.lines_starts[n];
    // (the env's row of the wall grid is asked for here, with the env's other rows: where it is used - once the agent's
    // position is known - it would be one more round trip in the chain position -> cell -> list -> walls)
    // (Unconditionally: ms_render points wg_geom / wg_starts / wg_pool_base at rows that exist when there is no grid, so that these
    // are part of the one batch of loads and not the body of a branch with a round trip of its own.)
    const float4 wg_geom_n = reinterpret_cast<const float4*>(sc.wg_geom)[n];
    const int wg_start_n = sc.wg_starts[n];
    const long long wg_pool_base_n = sc.wg_pool_base[n];                 // (its floorplan's vis lists: 64 bits - see MsScenery)
    float4* __restrict__ ln = reinterpret_cast<float4*>(sc.lines_vals) + base;
    const LineRows rows(ln, L);
    // --- every agent's heading and position, once per wave: lane i holds agent i (i < A <= 64; above that the
    // drawn lines fall back to drawn_line()).  sin/cos run in binary64, so they are worth sharing.
    float ag_s = 0.f, ag_c = 0.f;
    float2 ag_p = make_float2(0.f, 0.f);
    // (a lambda: a wave of several ray groups reads the agents afresh for every group rather than hold them in registers
    // through a group's epilogue - hot lines, and four registers the lighting cannot spare)
    auto load_agents = [&]() {
    ag_s = 0.f; ag_c = 0.f; ag_p = make_float2(0.f, 0.f);
    const int lane_a = n*A + min(lane, A - 1);   // (lanes past the last agent re-read it: loads without a guard overlap)
    if (ag.headings) {
        const float4 h = reinterpret_cast<const float4*>(ag.headings)[lane_a];
        const float angle = ag.angles[lane_a];
        const float2 p_ = reinterpret_cast<const float2*>(ag.positions)[lane_a];
        if (lane < A) {                                      // ms_physics' cache, valid while the angle has not changed
            ag_s = h.y; ag_c = h.z; ag_p = p_;
            if (f_bits(h.x) != f_bits(angle)) { const float2 sc_ = sincospi_called(angle/180.f); ag_s = sc_.x; ag_c = sc_.y; }   // (rare: a respawn)
        }
    } else if (lane < A) {
        if (out.workspace) {
            const float2 sc_ = reinterpret_cast<const float2*>(out.workspace + rc.ws_headings)[n*A + lane];
            ag_s = sc_.x; ag_c = sc_.y;
        } else {
            const float2 sc_ = sincospi_called(ag.angles[n*A + lane]/180.f);
            ag_s = sc_.x; ag_c = sc_.y;
        }
        ag_p = reinterpret_cast<const float2*>(ag.positions)[n*A + lane];
    }
    };
    if constexpr (STEP == 1) {
        // ---- the env's physics step (physics_kernel for its one agent, kernels.cu:179-230), every lane with the agent's state,
        // lane = wall for the tests.  Same functions, same operations in the same order as physics_kernel: the same bits.
        float2 p_ = reinterpret_cast<const float2*>(ag.positions)[n], v_ = reinterpret_cast<const float2*>(ag.velocity)[n];
        float w_ = ag.angvelocity[n], ang_ = ag.angles[n];
        // ---- what ms_step_physics runs around the step (MsStepExtras, MsMovement; physics_kernel<MOVE, EXTRA>), for this one agent,
        // statement for statement in its order; every lane works the same values out, lane 0 writes (uniform branches: a caller
        // that hands over neither pays two scalar compares)
        const MsStepExtras& ex = rc.ex;
        const MsMovement& mv = rc.mv;
        bool respawn_now = false;
        float2 spawn_p = make_float2(0.f, 0.f);
        float spawn_ang = 0.f;
        if (ex.respawn_mask || ex.lifespans) {                               // modules.py:361-366, :312-326
            bool reset = ex.respawn_mask && ex.respawn_mask[n];
            if (ex.lifespans) {
                int life = ex.lifespans[n] + 1;
                reset = reset | (life >= ex.max_lifespans[n]);
                const int fresh = ex.fresh_max[n];
                __builtin_amdgcn_wave_barrier();                                 // (every lane has read before lane 0 writes)
                if (lane == 0) {
                    if (reset) ex.max_lifespans[n] = fresh;
                    ex.lifespans[n] = reset ? 0 : life;
                    if (ex.respawn_mask) ex.respawn_mask[n] = reset ? 1 : 0;
                }
            }
            if (reset && ex.spawn_positions) {
                const long long c_ = min(max(ex.respawn_choice[n], 0ll), (long long)ex.n_spawns - 1);
                spawn_p = reinterpret_cast<const float2*>(ex.spawn_positions)[(size_t)n*ex.n_spawns + c_];
                spawn_ang = ex.spawn_angles[(size_t)n*ex.n_spawns + c_];
                respawn_now = true;
                if (!ex.respawn_after) {
                    p_ = spawn_p; ang_ = spawn_ang; v_ = make_float2(0.f, 0.f); w_ = 0.f;
                    if (lane == 0) {
                        reinterpret_cast<float2*>(ag.positions)[n] = p_;
                        ag.angles[n] = ang_;
                        reinterpret_cast<float2*>(ag.velocity)[n] = v_;
                        ag.angvelocity[n] = 0.f;
                    }
                }
            }
        }
        if (mv.actions) {                                                    // modules.py:57-66,106-118
            const long long act = min(max(mv.actions[n], 0ll), (long long)mv.n_actions - 1);
            const float dx = mv.table[3*act], dy = mv.table[3*act + 1], dw = mv.table[3*act + 2];
            const float a_ = 0.017453292519943295f*ang_;                     // np.pi/180*angles, in binary32 like torch
            const float s_ = sinf(a_), c_ = cosf(a_);
            const float gx = c_*dx - s_*dy, gy = s_*dx + c_*dy;
            if (mv.keep == 0.f) { w_ = dw; v_ = make_float2(gx, gy); }
            else { w_ = mv.keep*w_ + dw; v_ = make_float2(mv.keep*v_.x + gx, mv.keep*v_.y + gy); }
            if (lane == 0) {
                ag.angvelocity[n] = w_;
                reinterpret_cast<float2*>(ag.velocity)[n] = v_;
            }
        }
        const P2 p0 = p2(p_.x, p_.y);
        const P2 v0 = p2(v_.x, v_.y)/rc.fps;
        const float reach = wall_reach(p0, v0, agent_radius);
        const float reach2 = reach_squared(reach);
        const float my_reach = (reach == reach) ? reach : INFINITY;
        const float4 tk = make_float4(p0.x, p0.y, v0.x, v0.y);
        // the walls it can touch: the near list of its cell (the tier its reach asks for), or - outside the grid, faster than
        // the lists cover, crawling (wall_reach), no grid - every wall of the env
        bool listed_p = false;
        unsigned near_first = 0u;
        int n_src = max(L - AF, 0);
        if (rc.wg_cells_physics) {                                          // (uniform)
            const float4 geom = wg_geom_n;
            const float inv_cell = __builtin_amdgcn_rcpf(sc.wg_cell);
            const float fx = floorf((p0.x - geom.x)*inv_cell), fy = floorf((p0.y - geom.y)*inv_cell);
            const bool inside = (fx >= 0.f) & (fx < geom.z) & (fy >= 0.f) & (fy < geom.w);       // (NaNs: outside)
            const int cell_id = __builtin_amdgcn_readfirstlane(wg_start_n + (inside ? (int)fy*(int)geom.z + (int)fx : 0));
            const uint4 hdr = reinterpret_cast<const uint4*>(rc.wg_cells_physics)[cell_id];
            listed_p = __builtin_amdgcn_readfirstlane((inside & (my_reach <= sc.wg_reach)) ? 1 : 0) != 0;
            if (listed_p) {
                near_first = (unsigned)__builtin_amdgcn_readfirstlane((int)hdr.z);
                n_src = __builtin_amdgcn_readfirstlane((int)((my_reach <= sc.wg_reach_lo) ? (hdr.w & 0xffffu) : (hdr.w >> 16)));
            }
        }
        unsigned xb = f_bits(1.f);
        for (int k0 = 0; k0 < n_src; k0 += WAVE) {
            float4 u;
            if (listed_p) u = reinterpret_cast<const float4*>(sc.wg_near_rows)[near_first + (unsigned)min(k0 + lane, n_src - 1)];
            else u = rows.chunk(lane, AF + k0);
            if ((k0 + lane < n_src) && !wall_beyond(tk, u, reach2)) {
                const float xw = collision_cs(p2(tk.x, tk.y), p2(tk.z, tk.w), p2(u.x, u.y), p2(u.z, u.w), agent_radius);
                if (xw < 1.f) xb = min(xb, f_bits(xw));                      // (values in [+0, 1]: the bits order like the floats, as physics_kernel's atomicMin has it)
            }
        }
        #pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) xb = min(xb, (unsigned)__shfl_xor((int)xb, o, WAVE));
        const float x = bits_f(xb);
        // the epilogue, kernels.cu:224-227
        float2 p_new = make_float2(p_.x + x*v_.x/rc.fps, p_.y + x*v_.y/rc.fps);
        float turned = normalize_degrees(ang_ + x*w_/rc.fps);
        bool stopped = x < 1;
        if (stopped) { v_ = make_float2(0.f, 0.f); w_ = 0.f; }
        if (respawn_now && ex.respawn_after && ex.respawn_mask) {            // Explorer's order: the step first, then the new pose
            p_new = spawn_p; turned = spawn_ang;
            v_ = make_float2(0.f, 0.f); w_ = 0.f;
            stopped = true;
        }
        const float2 hsc = sincospi_called(turned/180.f);
        if (lane == 0) {
            reinterpret_cast<float2*>(ag.positions)[n] = p_new;
            ag.angles[n] = turned;
            if (ag.headings) reinterpret_cast<float4*>(ag.headings)[n] = make_float4(turned, hsc.x, hsc.y, 0.f);
            if (stopped) {
                reinterpret_cast<float2*>(ag.velocity)[n] = v_;
                ag.angvelocity[n] = w_;
            }
            rc.progress[n] = x;
        }
        if (ex.imu) {                                                        // modules.py:263-270, to_local_frame :24-31
            const float a_ = 0.017453292519943295f*turned;
            const float s_ = sinf(a_), c_ = cosf(a_);
            if (lane == 0) {
                ex.imu[3*n] = w_*ex.imu_ang_scale;                           // (the reciprocals: see physics_kernel)
                ex.imu[3*n + 1] = (c_*v_.x + s_*v_.y)*ex.imu_speed_scale;
                ex.imu[3*n + 2] = (-s_*v_.x + c_*v_.y)*ex.imu_speed_scale;
            }
        }
        ag_s = hsc.x; ag_c = hsc.y; ag_p = p_new;                            // the pose the rays are cast from
    } else {
        load_agents();
    }
    constexpr int AHEAD = MS_AHEAD;              // chunks of lines in flight (IMPL 2)

    // An agent's model line in world coordinates (draw_kernel, kernels.cu:297-318), from the cached heading where
    // there is one.  Lanes exchange data in here: call it from wave-uniform control flow only.
    // The model rows the two early users want, asked for up front and for every lane: row `lane` (the draw step's) and
    // row `lane mod M` (the first chunk's agent lines).  Fetched where they are used, behind those users' conditions,
    // they would drain the line chunks in flight.
    const float4 mdl_draw = reinterpret_cast<const float4*>(sc.model)[min(lane, sc.n_model - 1)];
    const float4 mdl_first = reinterpret_cast<const float4*>(sc.model)[lane - div_by(lane, rc.by_m)*sc.n_model];
    auto agent_line_m = [&](const int l_, const bool have_row, const float4 row) {
        const int l = min(max(l_, 0), AF - 1);
        if (A > WAVE) return drawn_line(sc, ag, n, l);
        const int la = div_by(l, rc.by_m);
        const float s_ = __shfl(ag_s, la, WAVE), c_ = __shfl(ag_c, la, WAVE);
        const float px_ = __shfl(ag_p.x, la, WAVE), py_ = __shfl(ag_p.y, la, WAVE);
        float4 mdl = row;
        if (!have_row) mdl = reinterpret_cast<const float4*>(sc.model)[l - la*sc.n_model];   // (uniform)
        float4 w;
        w.x = c_*mdl.x - s_*mdl.y + px_; w.y = s_*mdl.x + c_*mdl.y + py_;
        w.z = c_*mdl.z - s_*mdl.w + px_; w.w = s_*mdl.z + c_*mdl.w + py_;
        return w;
    };
    auto agent_line = [&](const int l_) { return agent_line_m(l_, false, make_float4(0.f, 0.f, 0.f, 0.f)); };
    // --- this wave's agent: heading and position (kernels.cu:334-339)
    float sn, cs;
    float2 pp;
    if (A <= WAVE) {
        sn = readlane_f(ag_s, a); cs = readlane_f(ag_c, a);
        pp = make_float2(readlane_f(ag_p.x, a), readlane_f(ag_p.y, a));
    } else {
        const float2 sc_ = sincospi_called(ag.angles[n*A + a]/180.f);
        sn = sc_.x; cs = sc_.y;
        pp = reinterpret_cast<const float2*>(ag.positions)[n*A + a];
    }
    PROBE_AT(1, pp.x)                                                    // the agents' state has arrived
    // --- the wall grid (MsScenery.wg_*, wallgrid_scan_kernel): the cell the agent stands in names the walls that can
    // matter to any ray cast from it; asked for here, as early as the position is known - the draw step and the ray
    // set-up below run while the answer travels.  No grid, or an agent outside it: every static wall (wg_count < 0).
    // (ms_render hands over wg_cells only when the grid holds for this call's near plane and field of view.)
    unsigned wg_first = 0u;
    int wg_count = -1;
    if constexpr (IMPL == 2) {
        if (sc.wg_cells) {                                                  // (the same for every wave of the launch)
            const float4 geom = wg_geom_n;
            const float inv_cell = __builtin_amdgcn_rcpf(sc.wg_cell);       // (cells are grown by a centimetre: an ulp is nothing)
            const float fx = floorf((pp.x - geom.x)*inv_cell), fy = floorf((pp.y - geom.y)*inv_cell);
            const bool inside = (fx >= 0.f) & (fx < geom.z) & (fy >= 0.f) & (fy < geom.w);    // (NaNs, an env without a grid: outside)
            // every wave reads a header that exists - its cell's, or the row at its env's start (the array is padded by one)
            const int cell_id = __builtin_amdgcn_readfirstlane(wg_start_n + (inside ? (int)fy*(int)geom.z + (int)fx : 0));
            const uint4 hdr = reinterpret_cast<const uint4*>(sc.wg_cells)[cell_id];
            wg_first = inside ? hdr.x : 0u;
            wg_count = inside ? (int)hdr.y : -1;
            if constexpr (NG > 1) {
                // (said to be the same in every lane - it is: one header, read by all - so that what is counted in the loops
                // over the list stays in scalar registers in the loops over the groups as well)
                wg_first = (unsigned)__builtin_amdgcn_readfirstlane((int)wg_first);
                wg_count = __builtin_amdgcn_readfirstlane(wg_count);
            }
        }
    }
    // --- draw: the wave of ray group 0 publishes its agent's model lines (kernels.cu:316-317).
    // Nobody reads them back from memory in this launch: every wave re-derives the agent lines it
    // needs (same inputs, same operations, same bits), so there is no cross-wave ordering to keep.
    if (r0 == 0) {
        for (int m0 = 0; m0 < sc.n_model; m0 += WAVE) {
            const float4 w = agent_line_m(a*sc.n_model + m0 + lane, m0 == 0, mdl_draw);
            if (m0 + lane < sc.n_model) ln[a*sc.n_model + m0 + lane] = w;
        }
    }
    // --- ray setup (kernels.cu:334-344)
    const float Rf = (float)R;
    auto ray_len = [&](const float rx_, const float ry_) {
#if MS_SQRT_NORMAL
        return sqrt_normal(rx_*rx_ + ry_*ry_);                          // (|r|^2 = (cos^2 + sin^2)(1 + uy^2): 1 to 1 + half_screen^2)
#else
        return sqrtf(rx_*rx_ + ry_*ry_);
#endif
    };
    auto ray_of = [&](const int r_, float& rx_, float& ry_, float& rlen_, float& near_) {
        // ray_y, kernels.cu:234-236.  (At a power-of-two resolution - 64, 128, 256, 512: every shape anyone runs - the
        // division by R only moves the exponent, and the product with 1/R is the correctly rounded quotient itself: one
        // multiply for the dozen instructions of a division.  The numerator is at least half_screen in size: no underflow.)
        const float num = (Rf - 2*(float)r_ - 1)*half_screen;
        const float uy = rc.inv_res != 0.f ? num*rc.inv_res : div_inrange(num, Rf);
        rx_ = cs*1.f - sn*uy; ry_ = sn*1.f + cs*uy;
        rlen_ = ray_len(rx_, ry_);
        near_ = div_inrange(agent_radius, rlen_);
    };
    float rx, ry, rlen, near;                                           // (the wave's first group's; the others' live in LDS)
    ray_of(r, rx, ry, rlen, near);

    // Screen-space bookkeeping for the culling below.  In the agent frame (x' forward, y' left) a point
    // is seen at screen coordinate ys = y'/x', i.e. at the continuous ray index c_a - ys*c_b (ray_y inverted).
    // Nothing with x' below x_clip can be hit: a hit has x' = s > agent_radius/|ru| > 2 x_clip.
    const float c_a = 0.5f*(Rf - 1.f), c_b = rc.c_b;                      // c_b = R/2/half_screen
    const float x_clip = rc.x_clip;                                        // agent_radius/2/sqrt(1 + half_screen^2)
    const float g0 = (float)r0;
    [[maybe_unused]] const int my_group = lane/GSIZE;

    // ---- the winner's loc and dot, recomputed from the same inputs (kernels.cu:356-364,374-375)
    // Everything the rest needs from memory about the winning line - its ends, its texel count and first texel - is
    // asked for here, for every lane, from a row that exists (the env's first for a miss): unconditional loads are
    // the ones hipcc lets overlap.
    // (Defined here, in front of the raycast, so that a wave of several ray groups can run them group by group from inside
    // it; they are CALLED at the wave's end, and it is there that they read their kernel arguments: see RenderArgs.)
    constexpr bool COLOUR = SHADE != 0;
    static_assert(COLOUR || OBS == 1, "without colour `screen` is NULL: the OBS instantiation");
    static_assert(OBS != 2 || (COLOUR && IMPL == 2 && STEP == 0), "no per-ray planes at all: a colour instantiation of the product raycast");
    constexpr bool PLANES = OBS != 2;            // (some per-ray plane may be wanted)
    // (is an optional output wanted?  colourless: a bit of the mask ms_render left in obs_subsample's upper half; else the pointer)
    // (round 6, MS_OBS_MASK: the colour instantiation with optional outputs - pooled RGB-D, crosshair ids - of ONE ray group a wave
    // asks the mask too: a whole env.step() at the headline shape 41.85 -> 41.56 us per step, two passes each way on one box; the
    // instantiation of four groups a wave does not - the Deathmatch shape's 512-ray render 150.6 -> 153.8 us with it: one more
    // live scalar in a kernel of 150 spilled ones (profiles/r06_ab_obs_mask.txt).  The plain one has no optional output to ask about.)
#ifndef MS_OBS_MASK
#define MS_OBS_MASK 1
#endif
    constexpr bool MASKED = !COLOUR || (MS_OBS_MASK != 0 && OBS >= 1 && IMPL == 2 && NG == 1);   // (the A/B builds' older raycasts ask the pointers)
    [[maybe_unused]] const int out_mask_ = MASKED ? (out.obs_subsample >> 8) : 0;
#define MS_WANTED(BIT, PTR) (MASKED ? ((out_mask_ & (BIT)) != 0) : ((PTR) != nullptr))
    // what depends on the winner's number alone: its row, its texel count and first texel
    // (plain scalars in and out: as a struct by value this cost every wave 32 bytes of scratch memory)
    auto winner_of = [&](const int nearest_idx, float4& hw_mem, int& tex_w, int& tstart) {
        const LateArgs late = late_args();       // (see RenderArgs: read where it is used, at the wave's end)
        // (uniform; constant-folded away in the colour instantiations)
        const bool want_texel_row = COLOUR || (OBS && MS_WANTED(OUT_SEEN, late->out.seen_stamp));
        const bool want_line = want_texel_row || MS_WANTED(OUT_LOCATIONS, late->out.locations) || MS_WANTED(OUT_DOTS, late->out.dots);
        const int row = min(max(nearest_idx, 0), max(L - 1, 0));
        hw_mem = make_float4(0.f, 0.f, 0.f, 0.f); tex_w = 1; tstart = 0;
        if (want_line) hw_mem = rows.row(row);
        if (want_texel_row) {
            const int* const l_tex_widths = late->sc.textures_widths;
            const int* const l_tex_starts = late->sc.textures_starts;
            tex_w = l_tex_widths[base + row]; tstart = l_tex_starts[base + row];
        }
    };
    // ... and the rest of a group's rays' lives: q = the group, r = this lane's ray of it, (rx, ry, rlen) = its direction
    auto finish_group = [&](const int q, const int r, const float rx, const float ry, const float rlen,
                            const float nearest_s, const int nearest_idx, const float4 hw_mem, const int tex_w, const int tstart) {
    const LateArgs late = late_args();       // (see RenderArgs: read where it is used, at the wave's end)
    // (uniform; constant-folded away in the colour instantiations)
    const bool want_texel_row = COLOUR || (OBS && MS_WANTED(OUT_SEEN, late->out.seen_stamp));
    const bool want_line = want_texel_row || MS_WANTED(OUT_LOCATIONS, late->out.locations) || MS_WANTED(OUT_DOTS, late->out.dots);
    float loc = NAN, dt = NAN;
    float4 hw = make_float4(0.f, 0.f, 0.f, 0.f);
    if (want_line) {
        float4 aw = make_float4(0.f, 0.f, 0.f, 0.f);
        if (__ballot((nearest_idx >= 0) & (nearest_idx < AF))) aw = agent_line(nearest_idx);
        if (nearest_idx >= 0) {
            hw = (nearest_idx < AF) ? aw : hw_mem;
            const float vx = hw.z - hw.x, vy = hw.w - hw.y;
            const float d = rx*vy - ry*vx;
            const float pqx = hw.x - pp.x, pqy = hw.y - pp.y;
            loc = div_inrange(pqx*ry - pqy*rx, d);
            const float dtop = rx*vx + ry*vy;
            const float dbot = rlen*sqrt_any(vx*vx + vy*vy);
            dt = div_inrange(dtop, dbot + 1.e-6f);
        }
    }
    [[maybe_unused]] const size_t o = ((size_t)n*A + a)*R + r;
    const float dist = nearest_s*rlen;
    if constexpr (PLANES) {
        if (!MASKED || (out_mask_ & (OUT_INDICES | OUT_LOCATIONS | OUT_DOTS | OUT_DISTANCES))) {      // (uniform)
            int* const o_indices = late->out.indices;
            float* const o_locations = late->out.locations;
            float* const o_dots = late->out.dots;
            float* const o_distances = late->out.distances;
            if (r < R) {
                if (!OBS || MS_WANTED(OUT_INDICES, o_indices)) MS_OUT_STORE(nearest_idx, &o_indices[o]);
                if (!OBS || MS_WANTED(OUT_LOCATIONS, o_locations)) MS_OUT_STORE(loc, &o_locations[o]);
                if (!OBS || MS_WANTED(OUT_DOTS, o_dots)) MS_OUT_STORE(dt, &o_dots[o]);
                if (!OBS || MS_WANTED(OUT_DISTANCES, o_distances)) MS_OUT_STORE(dist, &o_distances[o]);
            }
        }
    }

    // ---- pass 3: shade (kernels.cu:407-450)
    const bool is_hit = (nearest_idx >= 0) & (r < R);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    [[maybe_unused]] Filt f = Filt{0, 0, 0.f, 0.f};
    [[maybe_unused]] float intensity = 0.f;
    [[maybe_unused]] float tl0 = 0.f, tl1 = 0.f, tl2 = 0.f, tr0 = 0.f, tr1 = 0.f, tr2 = 0.f;
    if constexpr (COLOUR) {
        const float* const l_tex_vals = late->sc.textures_vals;
        const float* const l_baked = late->sc.baked_vals;
        const bool dynamic = is_hit & (nearest_idx < AF);
        // Rays that landed on an agent (dynamic) are lit from the lights (kernels.cu:432-436).  With a light grid
        // this wave does it here, on the LDS the raycast no longer needs; without one they leave black and their
        // ray group is queued for dynlight_kernel, launched right behind this kernel.
        [[maybe_unused]] unsigned light_telemetry = 0x80000000u;
#ifdef MS_NO_DYNLIGHT
        if (dynamic) intensity = 1.f;            // (an ablation: what would free dynamic lighting buy? the picture is wrong)
        if (false) {
#else
        if (__ballot(dynamic)) {
#endif
            if (sc.lg_vals) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const float cx_l = hw.x*(1 - loc) + hw.z*loc, cy_l = hw.y*(1 - loc) + hw.w*loc;   // kernels.cu:435
                const LightScene scl{sc.n_agents, sc.n_model, late->sc.lights_vals, late->sc.lights_widths, late->sc.lights_starts,
                                     late->sc.lg_vals, late->sc.lg_starts, late->sc.lg_geom, late->sc.lg_cell,
                                     late->sc.lg_list, late->sc.lg_pool, reinterpret_cast<const float4*>(late->sc.lg_pool_rows),
                                     rc.by_m.mul, rc.by_m.sh1, rc.by_m.sh2};   // (fetched now: see RenderArgs)
#if MS_PROBE
                unsigned lclk[2] = {0u, 0u};                     // (probe build: the lighting's own stamps - they take the places of the pair statistics)
                PROBE_VAL(14, (unsigned)clock64())
                intensity = grid_light_intensity(scl, ag, n, lane, dynamic, nearest_idx, cx_l, cy_l, L, ln,
                    reinterpret_cast<LightPair*>(&s_raw[wave][O_EPI]), reinterpret_cast<unsigned*>(&s_raw[wave][O_EPI + 2048]), light_telemetry, lclk);
                PROBE_VAL(12, lclk[0]) PROBE_VAL(13, lclk[1]) PROBE_VAL(11, (unsigned)clock64())
#else
                intensity = grid_light_intensity(scl, ag, n, lane, dynamic, nearest_idx, cx_l, cy_l, L, ln,
                    reinterpret_cast<LightPair*>(&s_raw[wave][O_EPI]), reinterpret_cast<unsigned*>(&s_raw[wave][O_EPI + 2048]), light_telemetry);
#endif
                PROBE_VAL(2, light_telemetry)
            } else if constexpr (NG == 1) {      // (wide waves never come here: without a light grid ms_render launches NG = 1 -
                // and `fan`, which nothing else needs, was one of the values the wide colour instantiations spilled)
                if (out.workspace && lane == 0) out.workspace[16 + atomicAdd(&out.workspace[0], 1)] = fan;
            }
        }
        // the 2-tap filter, and the texels and baked light under it - again for every lane (a miss looks at texel 0 of the
        // env's first line and throws the result away)
        PROBE_AT(5, tex_w)                                                   // the winner's line and texel row have arrived
        f = tex_filter(is_hit ? loc : 0.f, tex_w);
        const float bk_l = l_baked[tstart + f.l], bk_r = l_baked[tstart + f.r];
        const float* __restrict__ tl = l_tex_vals + 3*(size_t)(tstart + f.l);
        const float* __restrict__ tr = l_tex_vals + 3*(size_t)(tstart + f.r);
        tl0 = tl[0]; tl1 = tl[1]; tl2 = tl[2]; tr0 = tr[0]; tr1 = tr[1]; tr2 = tr[2];
        if (is_hit & !dynamic) intensity = f.lw*bk_l + f.rw*bk_r;
    }
    if constexpr (OBS >= 1) {
        if (MS_WANTED(OUT_SEEN, late->out.seen_stamp)) {           // explorer.py:34-58: which texels are seen for the first time
            bool fresh = false, fresh_last = false;
            const int last_env = sc.n_envs - 1;
            if (is_hit) {
                const float wf = (float)tex_w;
                const int along = (int)ms_min(floorf(wf*loc), wf - 1);      // explorer.py:38-41
                const int epoch = late->out.seen_epoch[n];
                // A look first: most texels in view were stamped frames ago, and an atomic that returns its old value
                // costs a round trip to the L2 per lane (a launch of nothing but stamped texels: 70 -> 39 us at 4096
                // envs x 256 rays).  Stamps only ever turn into the epoch during a launch, so a stale read can only
                // send a ray on to the exchange, where exactly one ray per texel sees the old stamp.
                if (late->out.seen_stamp[tstart + along] != epoch)
                    fresh = atomicExch(&late->out.seen_stamp[tstart + along], epoch) != epoch;
            } else if ((r < R) & (sc.n_texels_total > 0)) {
                // A ray that missed.  The reference gives it texel index -1 (explorer.py:36) and then sets `_seen[-1]`
                // (:47): the LAST texel of the whole scenery counts as seen from then on, to the credit of the last env,
                // whichever env's ray it was.  Kept as it is - a drop-in hands out the reference's rewards.
                const int last = sc.n_texels_total - 1;
                const int epoch = late->out.seen_epoch[last_env];
                if (late->out.seen_stamp[last] != epoch)
                    fresh_last = atomicExch(&late->out.seen_stamp[last], epoch) != epoch;
            }
            const unsigned long long fm = __ballot(fresh);
            if (fm && lane == 0) atomicAdd(&late->out.seen_count[n], __popcll(fm));
            if (__ballot(fresh_last) && lane == 0) atomicAdd(&late->out.seen_count[last_env], 1);
        }
        if (MS_WANTED(OUT_CENTRE, late->out.obs_centre)) {         // deathmatch.py:74-80: who is in the crosshair
            // (obs_subsample is a power of two - ms_render checks: shifts, not the thirty-instruction integer divisions a runtime
            // divisor costs here)
            const int sub = MASKED ? (late->out.obs_subsample & 0xff) : late->out.obs_subsample, sh = __builtin_ctz((unsigned)sub), W = R >> sh;
            const int r1 = (((W >> 1) - 1) << sh) + (sub >> 1), r2 = ((W >> 1) << sh) + (sub >> 1);
            if ((r == r1) | (r == r2)) {
                int seen = -1;
                if ((nearest_idx >= 0) & (nearest_idx < AF)) seen = div_by(nearest_idx, rc.by_m);
                late->out.obs_centre[((size_t)n*A + a)*2 + (r == r2 ? 1 : 0)] = seen;
            }
        }
    }

    if constexpr (COLOUR) {
        PROBE_AT(6, tl0)                                                     // ... its texels
        if (is_hit) {
            const float dn = 1 - dt*dt;
            s0 = dn*intensity*(f.lw*tl0 + f.rw*tr0);
            s1 = dn*intensity*(f.lw*tl1 + f.rw*tr1);
            s2 = dn*intensity*(f.lw*tl2 + f.rw*tr2);
        }
        if constexpr (PLANES) {
        float* const o_screen = late->out.screen;
        if (!OBS || MS_WANTED(OUT_SCREEN, o_screen)) {
            // stage RGB through LDS so the (R, 3) rows leave as three fully coalesced 256 B stores
            s_screen_w[3*lane] = s0; s_screen_w[3*lane + 1] = s1; s_screen_w[3*lane + 2] = s2;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int nfl = 3*min(n_live - q*WAVE, WAVE);
            float* __restrict__ scr = o_screen + 3*(((size_t)n*A + a)*R + r0 + q*WAVE);
            #pragma unroll
            for (int k = 0; k < 3; k++) {
                const int j = lane + k*WAVE;
                if (j < nfl) MS_OUT_STORE(s_screen_w[j], &scr[j]);
            }
        }
        }
    }
    // ---- pooled observations (modules.py:138-145,170-184,211-224): the mean over `sub` adjacent rays of the colour
    // and of the depth 1 - clamp((distance - agent_radius)/max_depth, 0, 1), summed pairwise across lanes
    if (OBS && ((COLOUR && MS_WANTED(OUT_RGB, late->out.obs_rgb)) || MS_WANTED(OUT_DEPTH, late->out.obs_depth))) {
        const int sub = MASKED ? (late->out.obs_subsample & 0xff) : late->out.obs_subsample;   // power of two, divides 64 and R (checked by the host)
        float p0 = s0, p1 = s1, p2 = s2;
        // (times the reciprocal ms_render left in the field, as ATen divides a tensor by a scalar - modules.py:176's
        // `(distances - agent_radius)/max_depth` on the device is `a * (1.f/b)`: torch's own bits, and ten instructions a group fewer
        // than the correctly rounded quotient this was until round 6)
        float pd = 1.f - ms_min(ms_max((dist - agent_radius)*late->out.obs_max_depth, 0.f), 1.f);
        // (the first two rounds - lanes 1 and 2 apart: all of them at the demo envs' four rays a pixel - stay inside quads of lanes:
        // DPP quad permutes, which ride on the add itself, instead of ds_bpermute's round trips through the LDS crossbar)
        auto quad_xor = [](const float v, auto ctrl) {
            return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), decltype(ctrl)::value, 0xf, 0xf, true));
        };
        constexpr std::integral_constant<int, 0xB1> XOR1{};      // quad_perm [1, 0, 3, 2]
        constexpr std::integral_constant<int, 0x4E> XOR2{};      // quad_perm [2, 3, 0, 1]
        if (sub > 1) {
            if constexpr (COLOUR) { p0 += quad_xor(p0, XOR1); p1 += quad_xor(p1, XOR1); p2 += quad_xor(p2, XOR1); }
            pd += quad_xor(pd, XOR1);
        }
        if (sub > 2) {
            if constexpr (COLOUR) { p0 += quad_xor(p0, XOR2); p1 += quad_xor(p1, XOR2); p2 += quad_xor(p2, XOR2); }
            pd += quad_xor(pd, XOR2);
        }
        for (int o2 = 4; o2 < sub; o2 <<= 1) {
            if constexpr (COLOUR) {
                p0 += __shfl_xor(p0, o2, WAVE); p1 += __shfl_xor(p1, o2, WAVE); p2 += __shfl_xor(p2, o2, WAVE);
            }
            pd += __shfl_xor(pd, o2, WAVE);
        }
        if (((lane & (sub - 1)) == 0) & (r < R)) {
            // (the mean: a sum over a power-of-two count - the host checks - divided by it, which only moves the exponent;
            // times the exact reciprocal is the same number for a twelfth of the instructions)
            // (... and sub, which divides R, a power of two as well: 1/sub is 2^-sh exactly, R/sub and r/sub are shifts - as
            // divisions by a run-time divisor these three were a hundred vector instructions per group of 64 rays, a seventh of
            // everything such a wave does: the pooled instantiation at 512 rays took 160.8 us where the plain one takes 138.7)
            const int sh = __builtin_ctz((unsigned)sub);
            const float inv = bits_f((uint32_t)(127 - sh) << 23);
            const int W = R >> sh, px = r >> sh;
            const size_t na = (size_t)n*A + a;
            if (COLOUR && MS_WANTED(OUT_RGB, late->out.obs_rgb)) {
                late->out.obs_rgb[(na*3 + 0)*W + px] = p0*inv;
                late->out.obs_rgb[(na*3 + 1)*W + px] = p1*inv;
                late->out.obs_rgb[(na*3 + 2)*W + px] = p2*inv;
            }
            if (MS_WANTED(OUT_DEPTH, late->out.obs_depth)) late->out.obs_depth[na*W + px] = pd*inv;
        }
    }
    };
    float nearest_s = INFINITY;
    int nearest_idx = -1;

#if MS_AB_IMPLS
    if constexpr (IMPL == 1) {
        // ------------------------------------------------------------------------------------------
        // (line, ray) pairs.  Pass 1 (lane = line) turns each line of the chunk into a conservative
        // INTEGER interval of this wave's rays; a DPP prefix sum lays all the intervals of the chunk end
        // to end, and pass 2 deals those (line, ray) pairs to the 64 lanes - every lane does one exact
        // intersection per step, whichever line and ray it belongs to.  Each hit is merged into its ray's
        // slot with a 64-bit LDS atomic min on the key (s bits << 32 | line): s > 0, so keys order by s,
        // ties by line index.  A second atomic keeps the runner-up.
        //
        // Why this equals the reference's order-dependent fold (kernels.cu:369-376): let (m, j*) be the
        // least key and m2 the runner-up's s.  If m < m2 - 1e-4f, then when the fold reaches j* its state is
        // inf or some s_k >= m2, so j* takes over, and nothing later can pass `s < m - 1e-4`: the fold ends
        // on (m, j*).  Otherwise the two best hits sit inside the hysteresis band (a ray through a shared
        // wall corner, coincident walls): with the third-best clearly behind, the fold of those two in line
        // order settles it; failing that the ray is re-done by the literal sequential fold below.
        // ------------------------------------------------------------------------------------------
        float4* const s_ray_w = reinterpret_cast<float4*>(&s_raw[wave][1024]);
        unsigned long long* const s_best_w = reinterpret_cast<unsigned long long*>(&s_raw[wave][2048]);
        unsigned long long* const s_second_w = reinterpret_cast<unsigned long long*>(&s_raw[wave][2560]);
        unsigned long long* const s_third_w = reinterpret_cast<unsigned long long*>(&s_raw[wave][3072]);
        int* const s_info_w = reinterpret_cast<int*>(&s_raw[wave][3584]);
        int* const s_mark_w = reinterpret_cast<int*>(&s_raw[wave][3840]);
        s_ray_w[lane] = make_float4(rx, ry, near, 0.f);
        s_best_w[lane] = ~0ull;
        s_second_w[lane] = ~0ull;
        s_third_w[lane] = ~0ull;
        const float last_local = (float)(r_last - r0);    // last live ray of this wave
        // pass 1 for one line (lane = line): the ray-independent half of the intersection into LDS, and the
        // conservative interval [lo, lo + len) of this wave's rays that can hit it
        auto line_setup = [&](const int c0, int& lo, int& len) {       // every lane comes in; dead ones leave with len 0
            const int l = c0 + lane;
            const bool live = l < L;
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live & (l >= AF)) w = ln[l];
            if (c0 < AF) {                                              // chunk with agent lines in it
                const float4 aw = agent_line(l);
                if (l < AF) w = aw;
            }
            const float pqx = w.x - pp.x, pqy = w.y - pp.y;            // PQ = Q - P
            const float dbx = w.z - pp.x, dby = w.w - pp.y;
            s_cand_w[lane] = Cand{pqx, pqy, w.z - w.x, w.w - w.y};  // v = b - a
            // agent-frame coordinates of both endpoints
            const float xa = __builtin_fmaf(cs, pqx, sn*pqy), ya = __builtin_fmaf(cs, pqy, -(sn*pqx));
            const float xb = __builtin_fmaf(cs, dbx, sn*dby), yb = __builtin_fmaf(cs, dby, -(sn*dbx));
            ray_interval<(MS_V1_OPTS & 1) ? 0 : 1>(xa, ya, xb, yb, live, x_clip, c_a, c_b, g0, last_local, lo, len);
        };
        int n_pairs_total = 0, n_windows = 0;    // telemetry

        for (int c0 = 0; c0 < L; c0 += WAVE) {
            int lo = 0, len = 0;
            line_setup(c0, lo, len);
            const int incl = wave_scan_add(len);
            const int first = incl - len;                                    // this line's first pair
            const int P = __builtin_amdgcn_readlane(incl, 63);
            PROBE_VAL(4, P)               // pairs in this chunk
            s_info_w[lane] = (first << 6) | (lo & 63);
            n_pairs_total += P; n_windows += (P + WAVE - 1)/WAVE;
            int carry = -1;
            for (int p0 = 0; p0 < P; p0 += WAVE) {
                // which line owns pair p0 + lane: lines mark their first pair, a max-scan spreads the marks
                s_mark_w[lane] = -1;
                if ((len > 0) & (first >= p0) & (first < p0 + WAVE)) s_mark_w[first - p0] = lane;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int owner = max(wave_scan_max(s_mark_w[lane]), carry);
                carry = __builtin_amdgcn_readlane(owner, 63);
                const int p = p0 + lane;
                const bool valid = p < P;
                const int j = valid ? owner : 0;
                const int info = s_info_w[j];
                const int rr = valid ? (info & 63) + (p - (info >> 6)) : 0;  // ray of this pair, wave-local
                const Cand cd = s_cand_w[j];
                const float4 ray = s_ray_w[rr];
                const float d = ray.x*cd.vy - ray.y*cd.vx;                   // cross(ru, v)
                const float nt = cd.pqx*ray.y - cd.pqy*ray.x;                // cross(PQ, ru)
                const float ad = fabsf(d);
                const float ntp = bits_f(f_bits(nt) ^ (f_bits(d) & 0x80000000u));
                // hit: 0 <= t <= 1 with t = nt/d  <=>  0 <= nt' <= |d| (exact, see light_blocked)
                const bool hit = valid & (ad >= 1.e-3f) & (ntp >= 0.f) & (ntp <= ad);
                if (hit) {
                    const float sv = (cd.pqx*cd.vy - cd.pqy*cd.vx)/d;        // q.s = cross(PQ, V)/UxV
                    if (ray.z < sv) {                                        // beyond the near plane, kernels.cu:369
                        const unsigned long long key = ((unsigned long long)f_bits(sv) << 32) | (unsigned)(c0 + j);
                        if constexpr ((MS_V1_OPTS & 2) != 0) {
                            // one atomic; the ray is flagged (its third slot, unused otherwise, set to 0) unless the
                            // loser of this merge is clearly behind the winner - see IMPL 2
                            const unsigned long long old = atomicMin(&s_best_w[rr], key);
                            const unsigned oh = (unsigned)(old >> 32);
                            if (oh != 0xffffffffu) {
                                const bool won = key < old;
                                const float so = bits_f(oh);
                                const float front = won ? sv : so, back = won ? so : sv;
                                if (!(front < back - 1.e-4f)) atomicMin(&s_third_w[rr], (unsigned long long)f_bits(front));
                            }
                        } else {
                        // keep the three smallest keys: whatever loses at one level drops to the next
                        const unsigned long long old1 = atomicMin(&s_best_w[rr], key);
                        const unsigned long long lose1 = old1 > key ? old1 : key;
                        if (lose1 != ~0ull) {
                            const unsigned long long old2 = atomicMin(&s_second_w[rr], lose1);
                            const unsigned long long lose2 = old2 > lose1 ? old2 : lose1;
                            if (lose2 != ~0ull) atomicMin(&s_third_w[rr], lose2);
                        }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            __builtin_amdgcn_wave_barrier();
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned long long best = s_best_w[lane], second = s_second_w[lane], third = s_third_w[lane];
        bool ambiguous = false;
        if (((MS_V1_OPTS & 2) != 0) && best != ~0ull) {
            nearest_s = bits_f((uint32_t)(best >> 32));
            nearest_idx = (int)(uint32_t)best;
            ambiguous = third == (best >> 32);
        } else if (best != ~0ull) {
            const float s1 = bits_f((uint32_t)(best >> 32)), s2 = bits_f((uint32_t)(second >> 32)), s3 = bits_f((uint32_t)(third >> 32));
            const int i1 = (int)(uint32_t)best, i2 = (int)(uint32_t)second;
            nearest_s = s1;
            nearest_idx = i1;
            if ((second != ~0ull) && !(s1 < s2 - 1.e-4f)) {
                // The two best hits are inside the band.  If every other hit is clearly behind both
                // (s2 < s3 - 1e-4f, s3 the third-smallest), no other line can interfere: each of the two
                // beats any state left by the others and the others never beat them, so the fold is the
                // fold of just these two in line order.
                if ((third == ~0ull) || (s2 < s3 - 1.e-4f)) {
                    const bool first_is_1 = i1 < i2;
                    const float sa = first_is_1 ? s1 : s2, sb = first_is_1 ? s2 : s1;
                    const int ia = first_is_1 ? i1 : i2, ib = first_is_1 ? i2 : i1;
                    const bool b_wins = sb < sa - 1.e-4f;
                    nearest_s = b_wins ? sb : sa;
                    nearest_idx = b_wins ? ib : ia;
                } else {
                    ambiguous = true;
                }
            }
        }
        // The literal fold for the rays that need it (kernels.cu:352-377).  Chunk by chunk, lane = line;
        // for each such ray the lanes compute that ray's hits on their lines, and the hits are folded in
        // line order into the ray's own state, which lives in the ray's lane.
        const unsigned long long amb = __ballot(ambiguous);
        // pair telemetry for tools/pair_stats.py - only on request (ms_debug_pair_telemetry): two atomics
        // per wave on one address are 1.3 ms at 262144 waves
        if (out.workspace && lane == 0 && rc.telemetry) {
            atomicAdd(&out.workspace[3], n_pairs_total); atomicAdd(&out.workspace[4], n_windows);
        }
        if (amb && out.workspace && lane == 0) {
            atomicAdd(&out.workspace[1], __popcll(amb));
            if (__popcll(amb) > 6) atomicAdd(&out.workspace[2], 1);
        }
        if (__popcll(amb) > 6) {
            // many such rays (a view full of coincident walls): every one of them walks the lines itself, lines
            // broadcast from LDS - but only the lines whose interval reaches one of these rays are looked at
            float x = INFINITY;
            int xi = -1;
            for (int c0 = 0; c0 < L; c0 += WAVE) {
                int lo = 0, len = 0;
                line_setup(c0, lo, len);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (unsigned long long todo = __ballot(len > 0); todo; todo &= todo - 1) {
                    const int j = __ffsll((long long)todo) - 1;
                    const int jlo = __builtin_amdgcn_readlane(lo, j), jhi = jlo + __builtin_amdgcn_readlane(len, j) - 1;
                    const unsigned long long span = ((jhi >= 63) ? ~0ull : ((2ull << jhi) - 1ull)) & ~((1ull << jlo) - 1ull);
                    if (!(span & amb)) continue;
                    if (ambiguous & (lane >= jlo) & (lane <= jhi)) {
                        const Cand cd = s_cand_w[j];
                        const float d = rx*cd.vy - ry*cd.vx;
                        const float nt = cd.pqx*ry - cd.pqy*rx;
                        const float ad = fabsf(d);
                        const float ntp = bits_f(f_bits(nt) ^ (f_bits(d) & 0x80000000u));
                        if ((ad >= 1.e-3f) & (ntp >= 0.f) & (ntp <= ad)) {
                            const float sv = (cd.pqx*cd.vy - cd.pqy*cd.vx)/d;
                            if ((near < sv) & (sv < x - 1.e-4f)) { x = sv; xi = c0 + j; }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (ambiguous) { nearest_s = x; nearest_idx = xi; }
        } else if (amb) {
            float x = INFINITY;
            int xi = -1;
            for (int c0 = 0; c0 < L; c0 += WAVE) {
                const int l = c0 + lane;
                float pqx = 0.f, pqy = 0.f, vx = 0.f, vy = 0.f;
                float4 aw = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c0 < AF) aw = agent_line(l);
                if (l < L) {
                    const float4 w = (l < AF) ? aw : ln[l];
                    pqx = w.x - pp.x; pqy = w.y - pp.y; vx = w.z - w.x; vy = w.w - w.y;
                }
                for (unsigned long long todo = amb; todo; todo &= todo - 1) {
                    const int jr = __ffsll((long long)todo) - 1;
                    const float jrx = readlane_f(rx, jr), jry = readlane_f(ry, jr), jnear = readlane_f(near, jr);
                    const float d = jrx*vy - jry*vx;
                    const float nt = pqx*jry - pqy*jrx;
                    const float ad = fabsf(d);
                    const float ntp = bits_f(f_bits(nt) ^ (f_bits(d) & 0x80000000u));
                    bool valid = false;
                    float sv = 0.f;
                    if ((l < L) & (ad >= 1.e-3f) & (ntp >= 0.f) & (ntp <= ad)) {
                        sv = (pqx*vy - pqy*vx)/d;
                        valid = jnear < sv;
                    }
                    unsigned long long m = __ballot(valid);
                    if (m) {
                        float xs = readlane_f(x, jr);
                        int xis = __builtin_amdgcn_readlane(xi, jr);
                        for (; m; m &= m - 1) {
                            const int j = __ffsll((long long)m) - 1;
                            const float sj = readlane_f(sv, j);
                            if (sj < xs - 1.e-4f) { xs = sj; xis = c0 + j; }
                        }
                        if (lane == jr) { x = xs; xi = xis; }
                    }
                }
            }
            if (ambiguous) { nearest_s = x; nearest_idx = xi; }
        }
    } else
#endif
    if constexpr (IMPL == 2) {
        // ------------------------------------------------------------------------------------------
        // (line, ray) pairs, second edition.  Same idea as IMPL 1 - pass 1 (lane = line) gives every line a
        // conservative integer interval of this wave's rays, pass 2 deals the (line, ray) pairs to the lanes - but
        //  * the lines that can be seen at all (about a third) are COMPACTED into an LDS list as the chunks go by,
        //    and pass 2 runs over the list when it fills up or the lines run out: full 64-pair windows instead of
        //    a ragged last window per chunk;
        //  * a line marks the bit of its first pair in an LDS bit vector; a window's 64 mark bits M are one
        //    broadcast read, and the line that owns pair q of the window is (#marks before the window) +
        //    popcount(M & bits 0..q) - 1: two mbcnt instructions instead of a marks array and a DPP max-scan;
        //  * ONE 64-bit LDS atomicMin per hit in the normal case.  Its return value is the ray's previous best, so
        //    the lane sees both parties of that merge; only when the loser is NEAR the winner (within 4e-4 of it:
        //    a few hits in a hundred) does it go on into the runner-up and third slots as in IMPL 1.  That is enough
        //    for IMPL 1's resolution to come out the same: every hit but the final best b loses exactly one merge,
        //    to a winner no nearer than b, so every hit within 3e-4 of s_b reaches the slots; the resolution only
        //    ever asks whether the runner-up is within 1e-4 of b and the third within 1e-4 of the runner-up, and
        //    whatever is missing from the slots is farther than that from either.
        //  * a line with an end behind the near clip plane is not clipped: its interval runs from the visible
        //    end's ray to the edge of the fan on the side it leaves by - the sign of cross(a, b).  (Clipping would
        //    only give less when the crossing is within centimetres of the agent.)
        // LDS per wave (V = V_CAP lines):  cand (V x 16 B) | info (V x 8 B: first ray - first pair, line) | ray (64 x 8 B: rx, ry)
        //               | near (64 x 4 B) | queue (128 x 2 B) | best, second, third (64 x 8 B each) | marks (4096 bits)
        // ------------------------------------------------------------------------------------------
        constexpr int V_CAP = VCAP, P_CAP = 4096 < 64*VCAP ? 4096 : 64*VCAP;
        // NG > 1 (several ray groups a wave, one after the other on ONE group's worth of per-ray state):
        //   cand (V x 16 B) | info (V x 8 B: first ray | rays << 16 of the line's interval among the span's rays, line) | queue
        //   | per group, O_EPI on: ray, near, best, second, third, marks as above | pinfo (V x 4 B: first ray - first pair of
        //   the lines that have pairs with this group) | gk (V x 2 B: which lines those are)        - 6912 B, six waves a SIMD
        constexpr int O_INFO = 16*V_CAP;
        constexpr int O_QUEUE = NG == 1 ? 24*V_CAP + 768 : 24*V_CAP;
        constexpr int O_RAY = NG == 1 ? 24*V_CAP : O_EPI, O_NEAR = O_RAY + 512, O_BEST = NG == 1 ? O_QUEUE + 256 : O_NEAR + 256;
        constexpr int O_MARK = O_BEST + 1536, O_PINFO = O_MARK + 512, O_GK = O_PINFO + 4*V_CAP;
        static_assert((NG == 1 ? O_MARK + 512 : O_GK + 2*V_CAP) == LDS_PER_WAVE && LDS_PER_WAVE >= 2560, "the LDS block: lists, rays, queue, three key slots, 4096 mark bits");
        static_assert(NG == 1 || O_EPI + 2560 <= LDS_PER_WAVE, "the epilogue's scratch fits the per-group region");
        int2* const s_info_w = reinterpret_cast<int2*>(&s_raw[wave][O_INFO]);
        // (direction and near plane in arrays of their own: at 8 and 4 bytes a ray, a window's reads - one ray per lane, the
        // rays mostly consecutive - touch every LDS bank once; as one 16-byte record per ray they were two-way conflicts)
        float2* const s_ray_w = reinterpret_cast<float2*>(&s_raw[wave][O_RAY]);
        float* const s_near_w = reinterpret_cast<float*>(&s_raw[wave][O_NEAR]);
        unsigned long long* const s_best_w = reinterpret_cast<unsigned long long*>(&s_raw[wave][O_BEST]);
        unsigned long long* const s_second_w = reinterpret_cast<unsigned long long*>(&s_raw[wave][O_BEST + 512]);
        unsigned long long* const s_third_w = reinterpret_cast<unsigned long long*>(&s_raw[wave][O_BEST + 1024]);
        unsigned* const s_mark_w = reinterpret_cast<unsigned*>(&s_raw[wave][O_MARK]);
        if constexpr (NG == 1) {
            s_ray_w[lane] = make_float2(rx, ry);
            s_near_w[lane] = near;
            s_best_w[lane] = ~0ull;
            s_mark_w[lane] = 0u; s_mark_w[lane + WAVE] = 0u;
            s_second_w[lane] = ~0ull;
            s_third_w[lane] = ~0ull;
        }
        // the run of directions of this wave's rays, from its rightmost ray (the last live one) to its leftmost (lane 0's of
        // the first group)                    (pseudo_angle with the reciprocal the hardware offers: the margin is 10^4 of its roundings wide)
        auto pseudo_angle_fast = [](const float x_, const float y_) {
            const float pq_ = y_*__builtin_amdgcn_rcpf(fabsf(x_) + fabsf(y_));
            return x_ < 0.f ? 2.f - pq_ : (pq_ < 0.f ? 4.f + pq_ : pq_);
        };
        const float pa_first = readlane_f(pseudo_angle_fast(rx, ry), 0);
        float pa_last = readlane_f(pseudo_angle_fast(rx, ry), min(n_live, WAVE) - 1);
        // the rays a list is made for - the wave's (NG = 1), or one span of its groups after the other: first ray, last live
        // ray counted from it, how many there are room for
        [[maybe_unused]] const float last_local = (float)(r_last - r0);
        float sp_g0 = g0, sp_last = (float)(r_last - r0), sp_nr = (float)WAVE;
        // pass 1 for one line (lane = line): the ray-independent half of the intersection, and the conservative
        // interval [lo, lo + len) of this wave's rays that can hit it
        // a chunk's lines as they are in memory, lane = line (dead lanes get the last row, agent rows whatever the last
        // render left there: neither is used)
        auto fetch = [&](const int c0) { return rows.chunk(lane, c0); };        // (not guarded: see w_first)
        // the full work on one line per lane (any line `l`; `agent_lines`: some lane holds one, wave-uniform): the
        // ray-independent half of the intersection, and the conservative interval [lo, lo + len) of this wave's rays
        // that can hit it.  Every lane comes in; dead ones leave with len 0
        auto line_math = [&](float4 w, const int l, const bool live, const bool agent_lines, const bool first_chunk, Cand& cd, int& lo, int& len) {
            if (agent_lines) {
                const float4 aw = agent_line_m(l, first_chunk, mdl_first);
                if (l < AF) w = aw;
            }
            const float pqx = w.x - pp.x, pqy = w.y - pp.y;            // PQ = Q - P
            const float dbx = w.z - pp.x, dby = w.w - pp.y;
            cd = Cand{pqx, pqy, w.z - w.x, w.w - w.y};                 // v = b - a
            float xa, ya, xb, yb;
            agent_frame(cs, sn, pqx, pqy, dbx, dby, xa, ya, xb, yb);
            ray_interval<(MS_V2_OPTS & 2) ? 1 : 0>(xa, ya, xb, yb, live, x_clip, c_a, c_b, sp_g0, sp_last, lo, len, sp_nr);
        };

        int n_pairs_total = 0, n_windows = 0;    // telemetry
        int n_list = 0, n_pairs = 0;             // lines and pairs in the list (wave-uniform)
        int n_drains = 0, list_n = 0;            // how often the list has been worked off, and how long it was the last time
        // pass 2 over the list: windows of 64 pairs
        auto drain = [&]() {
            n_drains++; list_n = n_list;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int before = 0;                      // marks in earlier windows = lines that start before this one
            n_pairs_total += n_pairs; n_windows += (n_pairs + WAVE - 1)/WAVE;
            if constexpr (MS_ABLATE == 2) n_pairs = 0;
            // the mark bits of all 64 possible windows in one read: lane w holds window w's, and each window fetches
            // its own with two v_readlane - no LDS round trip per window
            const unsigned long long my_marks = reinterpret_cast<const unsigned long long*>(s_mark_w)[lane];
            for (int p0 = 0; p0 < n_pairs; p0 += WAVE) {
                const unsigned mlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)my_marks, p0 >> 6);
                const unsigned mhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(my_marks >> 32), p0 >> 6);
                const unsigned long long M = ((unsigned long long)mhi << 32) | mlo;
                // marks at positions 0..lane of this window, via the bits of M >> 1 below the lane
                const unsigned long long Ms = M >> 1;
                const int upto = (int)(mlo & 1u) + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(Ms >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)Ms, 0u));
                const int p = p0 + lane;
                const bool valid = p < n_pairs;
                const int k = before + upto - 1;             // (past the last pair there are no marks: the last line, harmless)
                before += __popcll(M);
                const int2 info = s_info_w[k];
                const int rr = (p + info.x) & 63;            // ray of this pair, wave-local (in range as it is for valid pairs)
                const Cand cd = s_cand_w[k];
                const float2 ray = s_ray_w[rr];
                const float d = ray.x*cd.vy - ray.y*cd.vx;                   // cross(ru, v)
                const float nt = cd.pqx*ray.y - cd.pqy*ray.x;                // cross(PQ, ru)
                const float ad = fabsf(d);
                const float ntp = bits_f(f_bits(nt) ^ (f_bits(d) & 0x80000000u));
                // hit: 0 <= t <= 1 with t = nt/d  <=>  0 <= nt' <= |d| (exact, see light_blocked)
                const bool hit = valid & (ad >= 1.e-3f) & (ntp >= 0.f) & (ntp <= ad);
                if (hit) {
                    const float sv = div_inrange(cd.pqx*cd.vy - cd.pqy*cd.vx, d);        // q.s = cross(PQ, V)/UxV
                    const bool beyond = s_near_w[rr] < sv;                          // beyond the near plane, kernels.cu:369
                    if (beyond) {
                        const unsigned long long key = ((unsigned long long)f_bits(sv) << 32) | (unsigned)info.y;
                        const unsigned long long old = atomicMin(&s_best_w[rr], key);
                        const unsigned oh = (unsigned)(old >> 32);
                        if (oh != 0xffffffffu) {                             // there was a hit before: is the loser anywhere near?
                            const bool won = key < old;
                            const float so = bits_f(oh);
                            const float front = won ? sv : so, back = won ? so : sv;
                            if (back < front + 4.e-4f) {                     // rare: the loser may matter to the resolution
                                const unsigned long long lose1 = won ? old : key;
                                const unsigned long long old2 = atomicMin(&s_second_w[rr], lose1);
                                const unsigned long long lose2 = old2 > lose1 ? old2 : lose1;
                                if (lose2 != ~0ull) atomicMin(&s_third_w[rr], lose2);
                            }
                        }
                    }
                }
            }
            // the list starts over
            __builtin_amdgcn_wave_barrier();
            s_mark_w[lane] = 0u; s_mark_w[lane + WAVE] = 0u;
            n_list = 0; n_pairs = 0;
        };

        // a batch of up to 64 lines (lane = line `l`) into the list: interval, pair numbering, compaction
        auto admit = [&](const float4 w, const int l, const bool live, const bool agent_lines, const bool first_chunk) {
            Cand cd;
            int lo = 0, len = 0;
            line_math(w, l, live, agent_lines, first_chunk, cd, lo, len);
            const bool seen = len > 0;
            const unsigned long long vm = __ballot(seen);
            if (!vm) return;                                                 // uniform
            const int incl = wave_scan_add(len);
            const int chunk_pairs = __builtin_amdgcn_readlane(incl, 63);
            const int chunk_lines = __popcll(vm);
            if ((n_list + chunk_lines > V_CAP) | (n_pairs + chunk_pairs > P_CAP)) drain();
            if (seen) {
                const int k = n_list + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(vm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)vm, 0u));
                const int first = n_pairs + incl - len;                      // this line's first pair
                s_cand_w[k] = cd;
                s_info_w[k] = make_int2(lo - first, l);                      // pair p of the list is ray p + (lo - first)
                atomicOr(&s_mark_w[first >> 5], 1u << (first & 31));
            }
            n_list += chunk_lines; n_pairs += chunk_pairs;
            if constexpr ((MS_V2_OPTS & 1) != 0) drain();
        };

        // The lines this wave meets: the agents' lines (worked out from the agents' state, never read; the agent's own
        // are left out when the host has checked that its whole outline lies inside the near plane - MsScenery.model_radius
        // - as the reference's does: a hit on them is never `beyond`, kernels.cu:369, and seen from their middle they span
        // half the fan, a quarter of all the (line, ray) pairs a wave would test), then the walls - those on the cell's vis
        // list, or all of them.
        //
        // A list entry names a wall and the arc of directions it can be seen in from anywhere in the cell (wg_arc).  First
        // the entries are looked at on their own, 64 to an instruction: those whose arc misses the run of directions of
        // this wave's rays - most of them, the more so the narrower the wave's share of the field of view - are dropped,
        // the others' wall numbers queued in LDS (ballot + mbcnt).  Only queued walls have their rows fetched and go
        // through pass 1: a dozen instructions per 64 entries decide what used to cost ninety.  Items (agent lines, then
        // the queue) are worked through in batches of up to AHEAD chunks, whose rows are all in flight at once; the queue
        // holds Q_CAP walls, and a long list is a matter of several batches.  (All loads are unconditional: behind a branch
        // hipcc waits for every load in flight at the first use of any of them.)
        if constexpr (MS_ABLATE == 1) { if (out.indices) out.indices[(size_t)fan*WAVE + lane] = __float_as_int(near + rlen); return; }
        constexpr int Q_CAP = 128;
        unsigned short* const s_queue_w = reinterpret_cast<unsigned short*>(&s_raw[wave][O_QUEUE]);
        const bool listed = wg_count >= 0;                                   // (uniform)
        const int n_raw = listed ? wg_count : max(L - AF, 0);
        const int own0 = a*sc.n_model, own = rc.skip_own ? sc.n_model : 0;
        const int AL = AF - own;                                             // agent lines among the items
        const __amdgpu_buffer_rsrc_t list_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned*>(sc.wg_pool + wg_pool_base_n + wg_first), 0, listed ? 4*n_raw : 0, 0x00020000);
        auto raw = [&](const int k0) {                                       // entries k0 + lane of the list (past its end: 0)
            return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(list_rsrc, 4*(k0 + lane), 0, 0);
        };
        // the walk over the items for rays whose directions run from `pa_right` to `pa_left`; every batch of lines goes to
        // `admit_fn`, and `stop_fn` may call it off between batches
        auto walk = [&](const float pa_right, const float pa_left, auto&& admit_fn, auto&& stop_fn) {
            int wa8, wb8;
            wg_wedge(pa_right, pa_left, wa8, wb8);
            unsigned e_next[2] = {raw(0), raw(WAVE)};
            int raw_pos = 0, q_len = 0, al_left = AL;                            // (uniform)
            // (without a list the queue's entries are simply the env's walls in their order: what stands at its front is
            // counted here rather than read back from its 16-bit entries - an env may have more than 65536 walls, a list not)
            int q_first = 0;
            for (;;) {
                // fill the queue from the list
                while ((raw_pos < n_raw) & (q_len <= Q_CAP - WAVE)) {
                    const unsigned e = e_next[0];
                    e_next[0] = e_next[1];
                    e_next[1] = raw(raw_pos + 2*WAVE);
                    const int k = raw_pos + lane;
                    const int idx = listed ? (int)(e & 0xffffu) : k;
#ifndef MS_ARC_CULL
#define MS_ARC_CULL 1                                                            // (0: an A/B build that queues every listed wall)
#endif
                    const bool keep = (k < n_raw) & (!listed | !MS_ARC_CULL | wg_arcs_meet((int)((e >> 16) & 255u), (int)(e >> 24), wa8, wb8));
                    const unsigned long long km = __ballot(keep);
                    if (keep) s_queue_w[q_len + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u))] = (unsigned short)idx;
                    q_len += __popcll(km);
                    raw_pos += WAVE;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // this batch's items: what is left of the agents' lines, then the queue
                const int al0 = AL - al_left;                                    // the first agent-line item of this batch
                const int n_al = min(al_left, AHEAD*WAVE);
                const int n_items = n_al + min(q_len, AHEAD*WAVE - n_al);
                const int q_used = n_items - n_al;
                int l_it[AHEAD];
                float4 w_it[AHEAD];
                #pragma unroll
                for (int kk = 0; kk < AHEAD; kk++) {
                    const int i = kk*WAVE + lane;
                    const int ai = al0 + i;                                      // as an agent-line item
                    const int q_at = min(max(i - n_al, 0), Q_CAP - 1);
                    const int qe = listed ? (int)s_queue_w[q_at] : q_first + q_at;
                    l_it[kk] = (i < n_al) ? ai + (ai >= own0 ? own : 0) : AF + qe;
                    w_it[kk] = rows.load(l_it[kk]*16, 0);
                }
                PROBE_VAL(2, 0)                                                  // (slot 2: what the dynamic lighting had to do)
                PROBE_AT(3, w_it[0].x)                                           // ... the first chunk of rows
                #pragma unroll
                for (int kk = 0; kk < AHEAD; kk++) {
                    if (kk*WAVE >= n_items) continue;                            // uniform
                    admit_fn(w_it[kk], l_it[kk], kk*WAVE + lane < n_items, kk*WAVE < n_al, al0 + kk*WAVE == 0);
                }
                al_left -= n_al;
                // what the batch did not take of the queue moves to its front
                __builtin_amdgcn_wave_barrier();
                if (q_used < q_len) {
                    unsigned short keep_[Q_CAP/WAVE];
                    #pragma unroll
                    for (int j = 0; j < Q_CAP/WAVE; j++) keep_[j] = s_queue_w[min(q_used + j*WAVE + lane, Q_CAP - 1)];
                    __builtin_amdgcn_wave_barrier();
                    #pragma unroll
                    for (int j = 0; j < Q_CAP/WAVE; j++) if (j*WAVE + lane < q_len - q_used) s_queue_w[j*WAVE + lane] = keep_[j];
                }
                q_len -= q_used;
                q_first += q_used;
                if (((raw_pos >= n_raw) & (q_len == 0) & (al_left == 0)) || stop_fn()) break;
            }
        };
        if constexpr (NG == 1) {
            walk(pa_last, pa_first, admit, [] { return false; });
            if (n_pairs) drain();
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // The nearest hit of this lane's ray of the wave's group q from its three key slots, or - where they cannot tell - by
        // the literal fold.  (rx, ry, near: that ray's; the first group's are in registers, the others' come from LDS.)
        bool list_whole = true;                  // (NG > 1) the list in LDS is all the span's lines
        auto resolve_group = [&](const int q, const float rx, const float ry, const float near, float& nearest_s, int& nearest_idx) {
            const unsigned long long best = s_best_w[lane], second = s_second_w[lane], third = s_third_w[lane];
            bool ambiguous = false;
            if (best != ~0ull) {                     // the resolution of IMPL 1, word for word
                const float s1 = bits_f((uint32_t)(best >> 32)), s2 = bits_f((uint32_t)(second >> 32)), s3 = bits_f((uint32_t)(third >> 32));
                const int i1 = (int)(uint32_t)best, i2 = (int)(uint32_t)second;
                nearest_s = s1;
                nearest_idx = i1;
                if ((second != ~0ull) && !(s1 < s2 - 1.e-4f)) {
                    if ((third == ~0ull) || (s2 < s3 - 1.e-4f)) {
                        const bool first_is_1 = i1 < i2;
                        const float sa = first_is_1 ? s1 : s2, sb = first_is_1 ? s2 : s1;
                        const int ia = first_is_1 ? i1 : i2, ib = first_is_1 ? i2 : i1;
                        const bool b_wins = sb < sa - 1.e-4f;
                        nearest_s = b_wins ? sb : sa;
                        nearest_idx = b_wins ? ib : ia;
                    } else {
                        ambiguous = true;
                    }
                }
            }
            // The literal fold for the rays that need it (kernels.cu:352-377), as in IMPL 1
            const unsigned long long amb = __ballot(ambiguous);
            PROBE_VAL(11, n_pairs_total) PROBE_VAL(12, n_drains == 1 ? list_n : -1) PROBE_VAL(13, __popcll(amb))
    #if MS_PROBE
            const unsigned t_fold0 = (unsigned)clock64();
            PROBE_VAL(15, t_fold0)                                               // (with stamp 3: how long passes 1 and 2 took)
    #endif
            // pair telemetry for tools/pair_stats.py - only on request (ms_debug_pair_telemetry): two atomics
            // per wave on one address are 1.3 ms at 262144 waves
            if (out.workspace && lane == 0 && q == 0 && rc.telemetry) {
                atomicAdd(&out.workspace[3], n_pairs_total); atomicAdd(&out.workspace[4], n_windows);
            }
            if (amb && out.workspace && lane == 0) {
                atomicAdd(&out.workspace[1], __popcll(amb));
                if (__popcll(amb) > 6) atomicAdd(&out.workspace[2], 1);
            }
            if (amb && (NG == 1 ? n_drains == 1 : list_whole)) {
                // The usual case: the wave's list was worked off once, at the end, so all of it is still in LDS - every line
                // a ray of this wave can hit (the exact cull arguments above), ray-independent half of the intersection
                // ready, in LINE ORDER: the agents' lines in theirs, then the cell's vis list, which wallgrid_fill_kernel
                // writes in ascending wall number and the arc cull only thins.  The reference's fold (kernels.cu:352-377)
                // over the lines a ray does not hit is a no-op, so the literal fold over the list is the literal fold:
                // lane = ray, one broadcast LDS read per line, no memory traffic and no chain of dependent chunk loads
                // (the sweep over all the env's lines from memory below made such a wave the one its launch waited for:
                // 10-26 us against a mean life of 6-9; profiles/r04_probe_*.txt).
                const int rounds = (list_n + WAVE - 1)/WAVE;
                // (which of the two: lane = line costs ~55 instructions per ray and round of 64 lines, lane = ray ~25 per line -
                // measured in the probe build; a wave with seven such rays and a hundred lines once took the second: 9 us)
                if (2*(int)__popcll(amb)*rounds <= list_n + 16) {
                    // a few such rays (nearly always one or two): lane = line of the list, 64 at a time, read from LDS once; per
                    // ray every line's hit at once, then the ray's hits - a handful - folded in line order through a scalar
                    // loop into the ray's state, which lives in the ray's own lane
                    float x = INFINITY;
                    int xi = -1;
                    for (int k0 = 0; k0 < list_n; k0 += WAVE) {
                        const int k = min(k0 + lane, list_n - 1);
                        const Cand cd = s_cand_w[k];
                        const int line = s_info_w[k].y;
                        const float cpv = cd.pqx*cd.vy - cd.pqy*cd.vx;               // cross(PQ, V)
                        for (unsigned long long todo = amb; todo; todo &= todo - 1) {
                            const int jr = __ffsll((long long)todo) - 1;
                            const float jrx = readlane_f(rx, jr), jry = readlane_f(ry, jr), jnear = readlane_f(near, jr);
                            const float d = jrx*cd.vy - jry*cd.vx;
                            const float nt = cd.pqx*jry - cd.pqy*jrx;
                            const float ad = fabsf(d);
                            const float ntp = bits_f(f_bits(nt) ^ (f_bits(d) & 0x80000000u));
                            bool valid = false;
                            float sv = 0.f;
                            if ((k0 + lane < list_n) & (ad >= 1.e-3f) & (ntp >= 0.f) & (ntp <= ad)) {
                                sv = div_inrange(cpv, d);
                                valid = jnear < sv;
                            }
                            unsigned long long m = __ballot(valid);
                            if (m) {
                                float xs = readlane_f(x, jr);
                                int xis = __builtin_amdgcn_readlane(xi, jr);
                                for (; m; m &= m - 1) {
                                    const int j = __ffsll((long long)m) - 1;
                                    const float sj = readlane_f(sv, j);
                                    if (sj < xs - 1.e-4f) { xs = sj; xis = __builtin_amdgcn_readlane(line, j); }
                                }
                                if (lane == jr) { x = xs; xi = xis; }
                            }
                        }
                    }
                    if (ambiguous) { nearest_s = x; nearest_idx = xi; }
                } else {
                    // many (a view along a stack of coincident walls): lane = ray, every line of the list in turn
                    float x = INFINITY;
                    int xi = -1;
                    #pragma unroll 4
                    for (int k = 0; k < list_n; k++) {
                        const Cand cd = s_cand_w[k];
                        const int line = s_info_w[k].y;
                        if (ambiguous) {
                            const float d = rx*cd.vy - ry*cd.vx;
                            const float nt = cd.pqx*ry - cd.pqy*rx;
                            const float ad = fabsf(d);
                            const float ntp = bits_f(f_bits(nt) ^ (f_bits(d) & 0x80000000u));
                            if ((ad >= 1.e-3f) & (ntp >= 0.f) & (ntp <= ad)) {
                                const float sv = div_inrange(cd.pqx*cd.vy - cd.pqy*cd.vx, d);
                                if ((near < sv) & (sv < x - 1.e-4f)) { x = sv; xi = line; }
                            }
                        }
                    }
                    if (ambiguous) { nearest_s = x; nearest_idx = xi; }
                }
            } else
            if (__popcll(amb) > 6) {
                // many such rays (a view full of coincident walls): every one of them walks the lines itself, lines
                // broadcast from LDS - but only the lines whose interval reaches one of these rays are looked at
                float x = INFINITY;
                int xi = -1;
                for (int c0 = 0; c0 < L; c0 += WAVE) {
                    Cand mine;
                    int lo = 0, len = 0;
                    line_math(fetch(c0), c0 + lane, c0 + lane < L, c0 < AF, NG == 1 && c0 == 0, mine, lo, len);   // (NG > 1: the model row is read where it is needed, not held)
                    __builtin_amdgcn_wave_barrier();
                    s_cand_w[lane] = mine;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    for (unsigned long long todo = __ballot(len > 0); todo; todo &= todo - 1) {
                        const int j = __ffsll((long long)todo) - 1;
                        int jlo = __builtin_amdgcn_readlane(lo, j) - q*WAVE, jhi = jlo + __builtin_amdgcn_readlane(len, j) - 1;   // in this group's lanes
                        if ((jhi < 0) | (jlo >= WAVE)) continue;
                        jlo = max(jlo, 0);
                        const unsigned long long span = ((jhi >= 63) ? ~0ull : ((2ull << jhi) - 1ull)) & ~((1ull << jlo) - 1ull);
                        if (!(span & amb)) continue;
                        if (ambiguous & (lane >= jlo) & (lane <= jhi)) {
                            const Cand cd = s_cand_w[j];
                            const float d = rx*cd.vy - ry*cd.vx;
                            const float nt = cd.pqx*ry - cd.pqy*rx;
                            const float ad = fabsf(d);
                            const float ntp = bits_f(f_bits(nt) ^ (f_bits(d) & 0x80000000u));
                            if ((ad >= 1.e-3f) & (ntp >= 0.f) & (ntp <= ad)) {
                                const float sv = div_inrange(cd.pqx*cd.vy - cd.pqy*cd.vx, d);
                                if ((near < sv) & (sv < x - 1.e-4f)) { x = sv; xi = c0 + j; }
                            }
                        }
                    }
                }
                if (ambiguous) { nearest_s = x; nearest_idx = xi; }
            } else if (amb) {
                float x = INFINITY;
                int xi = -1;
                for (int c0 = 0; c0 < L; c0 += WAVE) {
                    const int l = c0 + lane;
                    float pqx = 0.f, pqy = 0.f, vx = 0.f, vy = 0.f;
                    float4 aw = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (c0 < AF) aw = agent_line(l);
                    if (l < L) {
                        float4 w = aw;                       // (not `l < AF ? aw : ln[l]`: hipcc turns that into a select of
                        if (l >= AF) w = ln[l];              //  two ADDRESSES and parks aw in scratch memory - for every wave)
                        pqx = w.x - pp.x; pqy = w.y - pp.y; vx = w.z - w.x; vy = w.w - w.y;
                    }
                    for (unsigned long long todo = amb; todo; todo &= todo - 1) {
                        const int jr = __ffsll((long long)todo) - 1;
                        const float jrx = readlane_f(rx, jr), jry = readlane_f(ry, jr), jnear = readlane_f(near, jr);
                        const float d = jrx*vy - jry*vx;
                        const float nt = pqx*jry - pqy*jrx;
                        const float ad = fabsf(d);
                        const float ntp = bits_f(f_bits(nt) ^ (f_bits(d) & 0x80000000u));
                        bool valid = false;
                        float sv = 0.f;
                        if ((l < L) & (ad >= 1.e-3f) & (ntp >= 0.f) & (ntp <= ad)) {
                            sv = div_inrange(pqx*vy - pqy*vx, d);
                            valid = jnear < sv;
                        }
                        unsigned long long m = __ballot(valid);
                        if (m) {
                            float xs = readlane_f(x, jr);
                            int xis = __builtin_amdgcn_readlane(xi, jr);
                            for (; m; m &= m - 1) {
                                const int j = __ffsll((long long)m) - 1;
                                const float sj = readlane_f(sv, j);
                                if (sj < xs - 1.e-4f) { xs = sj; xis = c0 + j; }
                            }
                            if (lane == jr) { x = xs; xi = xis; }
                        }
                    }
                }
                if (ambiguous) { nearest_s = x; nearest_idx = xi; }
            }
        };
        if constexpr (NG == 1) {
            resolve_group(0, rx, ry, near, nearest_s, nearest_idx);
        } else {
            // ------------------------------------------------------------------------------------------
            // Several ray groups a wave.  What an agent's waves each did for themselves - its state, its cell, the vis
            // list and its arc cull, the agents' lines, pass 1 on every line their wedges share - is done once, for a SPAN
            // of groups: pass 1 turns a line into an interval [lo, lo + len) of all the span's rays, and the lines that
            // have one go into the list with it.  Then group after group, on one group's worth of per-ray state: its rays
            // set up, the list's intervals clipped to its 64 rays and the (line, ray) pairs numbered (a prefix sum over the
            // list, 64 lines at a time; the lines that have pairs with the group are indexed), pass 2's windows, the
            // resolution - its literal fold over the whole list - and the epilogue, whose scratch sits behind the list.
            // A span is all the wave's groups; only if their lines do not fit the list (V_CAP) is it redone group by group,
            // the list then being worked off whenever it is full, as with NG = 1.
            // ------------------------------------------------------------------------------------------
            int* const s_pinfo_w = reinterpret_cast<int*>(&s_raw[wave][O_PINFO]);
            unsigned short* const s_gk_w = reinterpret_cast<unsigned short*>(&s_raw[wave][O_GK]);
            const int n_groups = (n_live + WAVE - 1)/WAVE;
            // pass 2 for the group whose rays are [lo_g, lo_g + 64) of the span's, over the list as it stands
            auto pass2_group = [&](const int lo_g) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                int base = 0, nj = 0;                 // pairs numbered, lines indexed so far (uniform)
                auto windows = [&]() {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    n_pairs_total += base; n_windows += (base + WAVE - 1)/WAVE;
                    const unsigned long long my_marks = reinterpret_cast<const unsigned long long*>(s_mark_w)[lane];
                    int before = 0;
                    for (int p0 = 0; p0 < base; p0 += WAVE) {
                        const unsigned mlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)my_marks, p0 >> 6);
                        const unsigned mhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(my_marks >> 32), p0 >> 6);
                        const unsigned long long M = ((unsigned long long)mhi << 32) | mlo;
                        const unsigned long long Ms = M >> 1;
                        const int upto = (int)(mlo & 1u) + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(Ms >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)Ms, 0u));
                        const int p = p0 + lane;
                        const bool valid = p < base;
                        const int j = before + upto - 1;             // (past the last pair there are no marks: the last line, harmless)
                        before += __popcll(M);
                        const int k = (int)s_gk_w[j];
                        const int rr = (p + s_pinfo_w[j]) & 63;      // ray of this pair, within the group
                        const Cand cd = s_cand_w[k];
                        const int line = s_info_w[k].y;
                        const float2 ray = s_ray_w[rr];
                        const float d = ray.x*cd.vy - ray.y*cd.vx;                   // cross(ru, v)
                        const float nt = cd.pqx*ray.y - cd.pqy*ray.x;                // cross(PQ, ru)
                        const float ad = fabsf(d);
                        const float ntp = bits_f(f_bits(nt) ^ (f_bits(d) & 0x80000000u));
                        const bool hit = valid & (ad >= 1.e-3f) & (ntp >= 0.f) & (ntp <= ad);
                        if (hit) {
                            const float sv = div_inrange(cd.pqx*cd.vy - cd.pqy*cd.vx, d);        // q.s = cross(PQ, V)/UxV
                            const bool beyond = s_near_w[rr] < sv;                   // beyond the near plane, kernels.cu:369
                            if (beyond) {
                                const unsigned long long key = ((unsigned long long)f_bits(sv) << 32) | (unsigned)line;
                                const unsigned long long old = atomicMin(&s_best_w[rr], key);
                                const unsigned oh = (unsigned)(old >> 32);
                                if (oh != 0xffffffffu) {
                                    const bool won = key < old;
                                    const float so = bits_f(oh);
                                    const float front = won ? sv : so, back = won ? so : sv;
                                    if (back < front + 4.e-4f) {
                                        const unsigned long long lose1 = won ? old : key;
                                        const unsigned long long old2 = atomicMin(&s_second_w[rr], lose1);
                                        const unsigned long long lose2 = old2 > lose1 ? old2 : lose1;
                                        if (lose2 != ~0ull) atomicMin(&s_third_w[rr], lose2);
                                    }
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    s_mark_w[lane] = 0u; s_mark_w[lane + WAVE] = 0u;
                    base = 0; nj = 0;
                };
                for (int k0 = 0; k0 < n_list; k0 += WAVE) {
                    const int k = k0 + lane;
                    const int iv = s_info_w[min(k, n_list - 1)].x;
                    const int lo = iv & 0xffff, hi = lo + (iv >> 16);
                    const int a0 = max(lo, lo_g), a1 = min(hi, lo_g + WAVE);
                    const int len_g = (k < n_list) ? max(a1 - a0, 0) : 0;
                    const unsigned long long vm = __ballot(len_g > 0);
                    if (!vm) continue;                                               // uniform
                    const int incl = wave_scan_add(len_g);
                    const int total = __builtin_amdgcn_readlane(incl, 63);
                    if (base + total > P_CAP) windows();                             // (64 lines x 64 rays always fit an empty numbering)
                    if (len_g > 0) {
                        const int j = nj + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(vm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)vm, 0u));
                        const int first = base + incl - len_g;
                        s_gk_w[j] = (unsigned short)k;
                        s_pinfo_w[j] = (a0 - lo_g) - first;                          // pair p of the numbering is ray p + this, of the group's
                        atomicOr(&s_mark_w[first >> 5], 1u << (first & 31));
                    }
                    base += total; nj += __popcll(vm);
                }
                if (base) windows();
            };
            // a group's rays into LDS, its slots and marks cleared
            auto setup_group = [&](const int q, float& qx, float& qy, float& ql, float& qn) {
                ray_of(r0 + q*WAVE + lane, qx, qy, ql, qn);
                __builtin_amdgcn_wave_barrier();                                     // (whoever read the region last is through)
                s_ray_w[lane] = make_float2(qx, qy);
                s_near_w[lane] = qn;
                s_best_w[lane] = ~0ull; s_second_w[lane] = ~0ull; s_third_w[lane] = ~0ull;
                s_mark_w[lane] = 0u; s_mark_w[lane + WAVE] = 0u;
            };
            // the span's rays and the run of directions they cover: from its last live ray (lane 1 works it out) to its first
            // (lane 0); returns the pseudo-angles in those two lanes
            auto span_of = [&](const int q0, const int sp_n) {
                const int sp_rays = min(n_live - q0*WAVE, sp_n*WAVE);
                sp_g0 = (float)(r0 + q0*WAVE); sp_last = (float)(sp_rays - 1); sp_nr = (float)(sp_n*WAVE);
                float wx, wy, wl_, wn_;
                ray_of(r0 + q0*WAVE + (lane == 0 ? 0 : sp_rays - 1), wx, wy, wl_, wn_);
                return pseudo_angle_fast(wx, wy);
            };
            // one group, once the list is what it is: pairs, nearest hits, everything behind them
            auto group_rest = [&](const int q, const int q_rel, const float gx, const float gy) {
                pass2_group(q_rel*WAVE);
                list_n = n_list;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                float ns = INFINITY;
                int ni = -1;
                resolve_group(q_rel, gx, gy, s_near_w[lane], ns, ni);
                LANE_AFRESH;
                float4 hw_mem; int tex_w, tstart;
                winner_of(ni, hw_mem, tex_w, tstart);
                finish_group(q, r0 + q*WAVE + lane, gx, gy, ray_len(gx, gy), ns, ni, hw_mem, tex_w, tstart);
                __builtin_amdgcn_wave_barrier();
            };
            // ---- all the wave's groups as one span
            bool overflow = false;
            {
                const float pa = span_of(0, n_groups);
                n_list = 0; list_whole = true;
                // a batch of lines into the list, each with its interval of the span's rays
                auto admit_shared = [&](const float4 w, const int l, const bool live, const bool agent_lines, const bool) {
                    if (overflow) return;
                    Cand cd;
                    int lo = 0, len = 0;
                    line_math(w, l, live, agent_lines, false, cd, lo, len);
                    const bool seen = len > 0;
                    const unsigned long long vm = __ballot(seen);
                    if (!vm) return;                                                 // uniform
                    const int chunk_lines = __popcll(vm);
                    if (n_list + chunk_lines > V_CAP) { overflow = true; return; }   // the span's lines do not fit: group by group, then
                    if (seen) {
                        const int k = n_list + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(vm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)vm, 0u));
                        s_cand_w[k] = cd;
                        s_info_w[k] = make_int2(lo | (len << 16), l);
                    }
                    n_list += chunk_lines;
                };
                walk(readlane_f(pa, 1), readlane_f(pa, 0), admit_shared, [&] { return overflow; });
            }
            if (!overflow) {
                #pragma unroll 1
                for (int q = 0; q < n_groups; q++) {
                    LANE_AFRESH;
                    if (q) load_agents();
                    float gx, gy, gl, gn;
                    setup_group(q, gx, gy, gl, gn);
                    group_rest(q, q, gx, gy);
                }
            } else {
                // ---- group by group (an env of dozens of agents, a cell with hundreds of walls in view): every group walks the
                // items for itself, and a list that fills up is worked off into the group's slots and started afresh
                #pragma unroll 1
                for (int q = 0; q < n_groups; q++) {
                    LANE_AFRESH;
                    if (q) load_agents();
                    const float pa = span_of(q, 1);
                    float gx, gy, gl, gn;
                    setup_group(q, gx, gy, gl, gn);
                    n_list = 0; list_whole = true;
                    auto admit_one = [&](const float4 w, const int l, const bool live, const bool agent_lines, const bool) {
                        Cand cd;
                        int lo = 0, len = 0;
                        line_math(w, l, live, agent_lines, false, cd, lo, len);
                        const bool seen = len > 0;
                        const unsigned long long vm = __ballot(seen);
                        if (!vm) return;                                             // uniform
                        const int chunk_lines = __popcll(vm);
                        if (n_list + chunk_lines > V_CAP) {
                            pass2_group(0);                                          // its slots hold what the list so far had to say
                            n_list = 0; list_whole = false;
                        }
                        if (seen) {
                            const int k = n_list + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(vm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)vm, 0u));
                            s_cand_w[k] = cd;
                            s_info_w[k] = make_int2(lo | (len << 16), l);
                        }
                        n_list += chunk_lines;
                    };
                    walk(readlane_f(pa, 1), readlane_f(pa, 0), admit_one, [] { return false; });
                    const float2 back = s_ray_w[lane];                               // (not held in registers through the walk)
                    group_rest(q, 0, back.x, back.y);
                }
            }
        }
    }
#if MS_AB_IMPLS
    else {
        for (int c0 = 0; c0 < L; c0 += WAVE) {
            // ---- pass 1: lane = line.  Each line of the chunk gets a CONSERVATIVE interval [r_lo, r_hi] of
            // continuous ray indices it can be hit from; a ballot per ray group turns those into one 64-bit
            // line mask per group.  Margins are ~1e3 rounding errors wide; anything doubtful is kept.
            const int l = c0 + lane;
            bool inc = false;
            float r_lo = 0.f, r_hi = 0.f, dmin2 = 0.f;
            // Depth bound per ray group: the largest squared hit distance any of its rays still holds.  The
            // fold state only ever decreases, so a line whose nearest point is beyond that bound can never
            // pass `s < nearest_s - 1e-4` for any ray of the group - now or later (exact, with 1e-4 slack).
            float bound2 = (nearest_idx >= 0) ? nearest_s*nearest_s*(rx*rx + ry*ry) : INFINITY;
            #pragma unroll
            for (int o = 1; o < GSIZE; o <<= 1) bound2 = fmaxf(bound2, __shfl_xor(bound2, o, WAVE));
            float4 aw = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c0 < AF) aw = agent_line(l);
            if (l < L) {
                const float4 w = (l < AF) ? aw : ln[l];
                const float pqx = w.x - pp.x, pqy = w.y - pp.y;            // PQ = Q - P
                const float dbx = w.z - pp.x, dby = w.w - pp.y;
                s_cand_w[lane] = Cand{pqx, pqy, w.z - w.x, w.w - w.y};  // v = b - a
                // agent-frame coordinates of both endpoints
                float xa = cs*pqx + sn*pqy, ya = cs*pqy - sn*pqx;
                float xb = cs*dbx + sn*dby, yb = cs*dby - sn*dbx;
                const bool fa = xa >= x_clip, fb = xb >= x_clip;
                inc = fa | fb | !(xa == xa) | !(xb == xb);                  // wholly behind the clip plane: never hit
                if (fa != fb) {                                             // clip the hidden end to x' = x_clip
                    const float t = (x_clip - xa)*__builtin_amdgcn_rcpf(xb - xa);
                    const float yc = ya + t*(yb - ya);
                    if (fa) { xb = x_clip; yb = yc; } else { xa = x_clip; ya = yc; }
                }
                const float ysa = ya*__builtin_amdgcn_rcpf(xa), ysb = yb*__builtin_amdgcn_rcpf(xb);
                const float ra = c_a - ysa*c_b, rb = c_a - ysb*c_b;
                const float marg = 0.05f + 1e-4f*(fabsf(ra) + fabsf(rb));
                r_lo = fminf(ra, rb) - marg - g0;
                r_hi = fmaxf(ra, rb) + marg - g0;
                // squared distance from the agent to the segment, shaved by 2e-4 so it is a lower bound
                const float vx = w.z - w.x, vy = w.w - w.y;
                float tc = -(pqx*vx + pqy*vy)*__builtin_amdgcn_rcpf(vx*vx + vy*vy);
                tc = fminf(fmaxf(tc, 0.f), 1.f);
                tc = (tc == tc) ? tc : 0.f;
                const float qx = pqx + tc*vx, qy = pqy + tc*vy;
                dmin2 = 0.9998f*(qx*qx + qy*qy);
            }
            unsigned long long my_mask = 0ull;
            #pragma unroll
            for (int k = 0; k < GROUPS; k++) {
                // excluded only if provably outside the group's rays [k*GSIZE, k*GSIZE + GSIZE - 1]; NaNs keep
                const float b2 = readlane_f(bound2, k*GSIZE);
                const bool ov = inc & !((r_lo > (float)(k*GSIZE + GSIZE - 1)) | (r_hi < (float)(k*GSIZE)) | (dmin2 > b2));
                const unsigned long long mk = __ballot(ov);
                if (my_group == k) my_mask = mk;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // ---- pass 2: lane = ray.  Every lane walks ITS group's lines in index order, so the fold is
            // the reference's sequential one (kernels.cu:352-377) minus lines that provably cannot hit.
            // hit: 0 <= t <= 1 with t = nt/d  <=>  0 <= nt' <= |d| (exact, see light_blocked).
            while (__ballot(my_mask != 0ull)) {
                const bool active = my_mask != 0ull;
                const int j = active ? __ffsll((long long)my_mask) - 1 : 0;
                my_mask &= my_mask - 1ull;
                const Cand cd = s_cand_w[j];
                const float d = rx*cd.vy - ry*cd.vx;                       // cross(ru, v)
                const float nt = cd.pqx*ry - cd.pqy*rx;                    // cross(PQ, ru)
                const float ad = fabsf(d);
                const float ntp = bits_f(f_bits(nt) ^ (f_bits(d) & 0x80000000u));
                const bool hit = active & (ad >= 1.e-3f) & (ntp >= 0.f) & (ntp <= ad);
                if (hit) {
                    const float sv = (cd.pqx*cd.vy - cd.pqy*cd.vx)/d;      // q.s = cross(PQ, V)/UxV
                    if ((near < sv) & (sv < nearest_s - 1.e-4f)) {
                        nearest_s = sv;
                        nearest_idx = c0 + j;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }

    }
#endif

    if constexpr (MS_ABLATE == 2 || MS_ABLATE == 3) { if (out.indices) out.indices[(size_t)fan*WAVE + lane] = nearest_idx + __float_as_int(nearest_s); return; }
    PROBE_AT(4, nearest_idx)                                             // the raycast is over
    if constexpr (NG == 1) {
        float4 hw_mem; int tex_w, tstart;
        winner_of(nearest_idx, hw_mem, tex_w, tstart);
        finish_group(0, r, rx, ry, rlen, nearest_s, nearest_idx, hw_mem, tex_w, tstart);
    }
    PROBE_DONE(fan)
#undef MS_WANTED
}

// ------------------------------------------------------------------------------------------------
// dynamic lighting of rays that hit an agent                               kernels.cu:432-436
// ------------------------------------------------------------------------------------------------
// Second launch of ms_render: ONE WORKGROUP PER (env, agent, 64-ray group), kept out of render_kernel
// so that kernel stays at 64 VGPRs.  Every wave of the workgroup reads the group's 64 hit indices and
// the workgroup leaves at once unless one of them is an agent line (~1 group in 7 on the benchmark
// workload).  Such rays need light_intensity() at the hit point: lights x walls occlusion tests per
// ray.  That is done cooperatively and exactly:
//   * all four waves hold the same per-ray state (lane = ray) and split the WALLS between them.
//   * per target agent, lights are ranked NEAREST FIRST.  With every intensity >= 0 the sum
//     0.1 + sum_i 2 I_i / max(d_i^2, 1) over unblocked lights only grows, so once the lights proven
//     unblocked so far add up to >= 1.001 the reference's min(sum, 1) is exactly 1 whatever the
//     remaining lights do (the 1e-3 dwarfs the reordering error of a <= 64-term float sum), and that
//     ray is done.  Phase 1 evaluates the four nearest lights (usually the target's own room light
//     settles it); phase 2 the remaining ones.  A ray that never saturates has every light evaluated
//     and is summed in the reference's light order.  In both phases the waves split the WALLS.
//   * within a wave, lane = wall: a wall can only shadow the target from a light if it reaches into
//     the CORRIDOR light -> target (a box around that segment grown by the extent of the hit points);
//     surviving (wall, light) pairs are compacted into the wave's LDS pair list.
//   * lane = pair, loop over the open rays: the reference's obstructed() test; a hit ORs the light's
//     bit into that ray's shadow words (LDS atomic, shared by the four waves).

__global__ __launch_bounds__(WG) void dynlight_kernel(
        const MsScenery sc, const MsAgents ag, const MsRender out, const int R) {
    __shared__ LightPair s_pair[WAVES][PAIRS];
    __shared__ unsigned s_shadow[2*WAVE];        // per ray: 64 light bits, OR-ed by all waves

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int fan = blockIdx.x;
    if (out.workspace) {                         // compact list from render_kernel: the busy groups start first
        if (fan >= out.workspace[0]) return;
        fan = out.workspace[16 + fan];
    }
    const int A = sc.n_agents, AF = sc.n_agents*sc.n_model;
    const int G = (R + WAVE - 1)/WAVE, F = A*G;
    const int n = fan / F, rem = fan - n*F, a = rem / G, g = rem - a*G;
    const int r = g*WAVE + lane;
    const size_t o = ((size_t)n*A + a)*R + r;
    // everything that does not depend on the indices is requested before they are looked at
    const int L = sc.lines_widths[n];
    const int base = sc.lines_starts[n];
    const int num_i = sc.lights_widths[n];
    const int lbase = sc.lights_starts[n];
    int nearest_idx = -1;
    float loc = 0.f, dt = 0.f;
    if (r < R) { nearest_idx = out.indices[o]; loc = out.locations[o]; dt = out.dots[o]; }
    const bool dynamic = (nearest_idx >= 0) & (nearest_idx < AF);
    const unsigned long long dyn = __ballot(dynamic);
    if (!dyn) return;                            // uniform across the workgroup: every wave sees the same 64 rays

    const float4* __restrict__ ln = reinterpret_cast<const float4*>(sc.lines_vals) + base;
    const float* __restrict__ lights = sc.lights_vals + 3*(size_t)lbase;
    float4 hw = make_float4(0.f, 0.f, 0.f, 0.f);
    Filt f = Filt{0, 0, 0.f, 0.f};
    float t0[3] = {0.f, 0.f, 0.f}, t1[3] = {0.f, 0.f, 0.f};
    if (dynamic) {
        hw = ln[nearest_idx];                               // the agent line render_kernel drew and published (kernels.cu:316-317)
        const int start = base + nearest_idx;
        f = tex_filter(loc, sc.textures_widths[start]);
        const int tstart = sc.textures_starts[start];
        const float* __restrict__ tl = sc.textures_vals + 3*(size_t)(tstart + f.l);
        const float* __restrict__ tr = sc.textures_vals + 3*(size_t)(tstart + f.r);
        #pragma unroll
        for (int k = 0; k < 3; k++) { t0[k] = tl[k]; t1[k] = tr[k]; }
    }
    const float cx_l = hw.x*(1 - loc) + hw.z*loc, cy_l = hw.y*(1 - loc) + hw.w*loc;   // kernels.cu:435
    const int my_target = dynamic ? nearest_idx / sc.n_model : -1;

    float acc = AMBIENT;                 // the reference's in-order sum, for rays that do not saturate
    bool saturated = false;
    for (int i0 = 0; i0 < num_i; i0 += WAVE) {
        const int ni = min(WAVE, num_i - i0);
        // lane i holds light i0+i
        float Ix = 0.f, Iy = 0.f, Ii = 0.f;
        if (lane < ni) { Ix = lights[3*(i0 + lane)]; Iy = lights[3*(i0 + lane) + 1]; Ii = lights[3*(i0 + lane) + 2]; }
        // the shortcut needs non-negative, finite contributions and all lights in this one group
        const bool shortcut = (num_i <= WAVE) & (__ballot((lane < ni) & !(Ii >= 0.f)) == 0ull);
        __syncthreads();
        if (wave == 0) { s_shadow[2*lane] = 0u; s_shadow[2*lane + 1] = 0u; }
        unsigned long long todo = dyn;
        while (todo) {                                       // uniform across the workgroup
            const int j = __ffsll((long long)todo) - 1;
            const int target = __builtin_amdgcn_readlane(my_target, j);
            const bool mine = dynamic & (my_target == target);
            todo &= ~__ballot(mine);
            const float2 T = reinterpret_cast<const float2*>(ag.positions)[n*A + target];
            // extent of the hit points around the target, + float slack
            float rho = mine ? sqrtf((cx_l - T.x)*(cx_l - T.x) + (cy_l - T.y)*(cy_l - T.y)) : 0.f;
            rho = wave_max_f(rho) + 2e-3f + 1e-4f*(fabsf(T.x) + fabsf(T.y));
            // corridor frame of light `lane`: unit vector e from the light to the target, length el
            const float dx = T.x - Ix, dy = T.y - Iy;
            const float key = dx*dx + dy*dy;
            const float el = sqrtf(key);
            const float ex = dx/el, ey = dy/el;
            // rank the lights by distance to the target (ties by slot); every wave computes the same
            int rank = 0;
            for (int q = 0; q < ni; q++) {
                const float kq = readlane_f(key, q);
                rank += ((kq < key) | ((kq == key) & (q < lane))) ? 1 : 0;
            }
            if (lane >= ni) rank = -1;
            // the light ranked oi: the lane whose rank is oi
            auto ranked = [&](int oi) { return __ffsll((long long)__ballot(rank == oi)) - 1; };

            // sweep: evaluates the lights ranked [o_lo, o_hi) for the open rays of this target against this
            // wave's share of the walls (chunk `wave`, `wave + 4`, ...: few dependent loads per wave)
            auto sweep = [&](int o_lo, int o_hi, unsigned long long open) {
                int cnt = 0;
                auto flush = [&]() {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    for (int p0 = 0; p0 < cnt; p0 += WAVE) {
                        const LightPair pr = s_pair[wave][min(p0 + lane, cnt - 1)];
                        const P2 I = p2(pr.ix, pr.iy);
                        for (unsigned long long rays = open; rays; rays &= rays - 1) {
                            const int jr = __ffsll((long long)rays) - 1;
                            const P2 C = p2(readlane_f(cx_l, jr), readlane_f(cy_l, jr));
                            if ((p0 + lane < cnt) && light_blocked(I, C - I, pr.ax, pr.ay, pr.vx, pr.vy))
                                atomicOr(&s_shadow[2*jr + (pr.light >> 5)], 1u << (pr.light & 31));
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    cnt = 0;
                };
                const int first = AF + wave*WAVE;
                float4 wn = make_float4(0.f, 0.f, 0.f, 0.f);
                if (first + lane < L) wn = ln[first + lane];
                for (int l0 = first; l0 < L; l0 += WAVES*WAVE) {
                    const bool live = l0 + lane < L;
                    const float4 w = wn;
                    if (l0 + WAVES*WAVE + lane < L) wn = ln[l0 + WAVES*WAVE + lane];      // next chunk in flight
                    // wall relative to the target, and its margin
                    const float ax = w.x - T.x, ay = w.y - T.y, bx = w.z - T.x, by = w.w - T.y;
                    const float m = rho + 1e-4f*(fabsf(ax) + fabsf(ay) + fabsf(bx) + fabsf(by));
                    for (int oi = o_lo; oi < o_hi; oi++) {
                        const int i = ranked(oi);
                        const float cex = readlane_f(ex, i), cey = readlane_f(ey, i), cel = readlane_f(el, i);
                        // coordinates along / across the corridor, origin at the target, light at -cel
                        const float ua = cex*ax + cey*ay, va = cex*ay - cey*ax;
                        const float ub = cex*bx + cey*by, vb = cex*by - cey*bx;
                        const bool outside = ((ua > m) & (ub > m)) | ((ua < -cel - m) & (ub < -cel - m)) |
                                             ((va > m) & (vb > m)) | ((va < -m) & (vb < -m));
                        const bool keep = live & !outside;
                        const unsigned long long km = __ballot(keep);
                        if (km) {
                            const int nk = __popcll(km);
                            if (cnt + nk > PAIRS) flush();
                            if (keep) s_pair[wave][cnt + __popcll(km & ((1ull << lane) - 1ull))] =
                                LightPair{w.x, w.y, w.z - w.x, w.w - w.y, readlane_f(Ix, i), readlane_f(Iy, i), i, 0};
                            cnt += nk;
                        }
                    }
                }
                if (cnt) flush();
            };

            // phase 1: the NEAR_LIGHTS nearest lights
            constexpr int NEAR_LIGHTS = 4;
            const int n1 = min(NEAR_LIGHTS, ni);
            unsigned long long open = __ballot(mine);
            sweep(0, n1, open);
            __syncthreads();
            int near_i[NEAR_LIGHTS];                         // (ballots must run with every lane active)
            #pragma unroll
            for (int oi = 0; oi < NEAR_LIGHTS; oi++) near_i[oi] = ranked(min(oi, n1 - 1));
            if (mine) {
                const unsigned long long blocked = ((unsigned long long)s_shadow[2*lane + 1] << 32) | s_shadow[2*lane];
                float part = AMBIENT;                        // order-free sum of the lights proven unblocked
                #pragma unroll
                for (int oi = 0; oi < NEAR_LIGHTS; oi++) {
                    const int i = near_i[oi];
                    if ((oi < n1) && !((blocked >> i) & 1ull)) {
                        const float d2 = len2(p2(readlane_f(Ix, i), readlane_f(Iy, i)) - p2(cx_l, cy_l));
                        part += LUMINANCE*readlane_f(Ii, i)/ms_max(d2, 1.f);
                    }
                }
                saturated = shortcut & (part >= 1.001f);
            }
            // phase 2: whatever is left, all the remaining lights
            open = __ballot(mine & !saturated);              // identical in every wave
            if (open && ni > n1) sweep(n1, ni, open);
        }
        __syncthreads();
        if (!__ballot(dynamic & !saturated)) break;                  // every ray clamps to 1: no sum needed
        const unsigned long long blocked = ((unsigned long long)s_shadow[2*lane + 1] << 32) | s_shadow[2*lane];
        for (int i = 0; i < ni; i++) {                               // kernels.cu:261-264, in light order
            const P2 I = p2(readlane_f(Ix, i), readlane_f(Iy, i));
            const float d2 = len2(I - p2(cx_l, cy_l));
            if (!((blocked >> i) & 1ull)) acc += LUMINANCE*readlane_f(Ii, i)/ms_max(d2, 1.f);
        }
    }
    if (dynamic & (wave == 0)) {                             // kernels.cu:441-445
        const float intensity = saturated ? 1.f : ms_min(acc, 1.f);
        const float dn = 1 - dt*dt;
        out.screen[3*o]     = dn*intensity*(f.lw*t0[0] + f.rw*t1[0]);
        out.screen[3*o + 1] = dn*intensity*(f.lw*t0[1] + f.rw*t1[1]);
        out.screen[3*o + 2] = dn*intensity*(f.lw*t0[2] + f.rw*t1[2]);
    }
}
