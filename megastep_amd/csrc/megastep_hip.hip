// megastep_hip.hip -- gfx950 (MI355X / CDNA4) simulation core behind include/megastep_hip.h.
//
// Eleven kernels, all written wave64-first (DESIGN.md section 3 has the full story of each):
//
//   physics_kernel<MOVE, EXTRA>   one wavefront per env: lane = agent for the state, the reach and the agent-agent
//                   tests; the walls from the near lists of the agents' cells of the wall grid (rows of the dozen walls
//                   within reach, dealt to the lanes one (wall, agent) pair each) - or, where no list applies, a sweep
//                   over all the env's walls (buffer loads in flight, lane = wall, reach boxes in scalar registers,
//                   compacted pairs); the exact collision test, atomicMin fold, integration epilogue; leaves each
//                   agent's sin/cos for the renderer.  MOVE = 1 runs the movement modules' velocity update first,
//                   EXTRA = 1 the envs' respawn / lifespan / IMU bookkeeping.
//                                                            (reference: kernels.cu:179-230, modules.py:24-118,263-366)
//   render_kernel<IMPL, RW, OBS>   one wavefront per (env, agent, 64-ray group).  The lines it meets: the other agents'
//                   and the walls on the vis list of the agent's cell of the wall grid, less those whose view arc misses
//                   the wave's rays.  Pass 1 (lane = line) turns every line into a conservative interval of the wave's
//                   rays and compacts the visible ones into an LDS list; pass 2 deals the (line, ray) pairs of the list
//                   to the lanes, 64 at a time, one exact intersection each, merged per ray with one 64-bit LDS atomic;
//                   the order-dependent nearest-hit rule is resolved from the three smallest keys (or a literal fold
//                   where it must be); rays that landed on an agent are lit through the light grid; shading; optional
//                   pooled observations, crosshair ids and first-sight books.  draw, raycast and shader (three launches +
//                   five allocations in the reference) are one launch.  IMPL = 2 is the product; 1 (per-chunk pair
//                   windows) and 0 (literal order, every line) exist in -DMS_AB_IMPLS=1 builds for A/B runs and produce
//                   the same bits.                               (reference: kernels.cu:297-475)
//   render_prep_kernel, dynlight_kernel   the renderer's helpers for callers without a heading cache / light grid.
//   visibility_kernel, bake_sum_kernel    the two-phase bake: per (representative env, light) the walls that can shadow
//                   each angular bin; per texel the sum over the lights, occluders looked up by bin.
//   bake_kernel     the one-pass bake: one workgroup per env, lane = texel, the env's occluders staged once in LDS.
//                                                            (reference: kernels.cu:238-293)
//   lightgrid_kernel, lightlist_kernel   per (cell, light) LIT / DARK / UNKNOWN verdicts and candidate walls.
//   wallgrid_scan_kernel, wallgrid_fill_kernel   the wall grid: per cell of a floorplan which walls can matter to a ray
//                   from the cell (one wall hiding another from the whole cell, exactly) and which an agent in it can
//                   touch; and the lists made of that.   (replaces the all-lines loops kernels.cu:203-205,352-377)
//
// Numerics contract: IEEE binary32 evaluated as the reference source is written -- compiled with
// -ffp-contract=off, correctly rounded divide/sqrt, no fast-math -- so that collision masks and hit
// indices are bit-identical to the CPU oracle and floats agree far inside the 1e-5 tolerance.
// Every shortcut below (division-free tests, culling, hoisting) is exact, not approximate.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/megastep_hip_test.h"

namespace {

constexpr float AMBIENT = .1f;      // kernels.cu:9
constexpr float LUMINANCE = 2.f;    // kernels.cu:240
constexpr int WAVE = 64;
constexpr int WG = 256;             // 4 waves per workgroup
constexpr int WAVES = WG/WAVE;

thread_local int g_last_hip_error = 0;
int g_pair_telemetry = 0;              // ms_debug_pair_telemetry
int g_ray_groups = 0;                  // ms_debug_ray_groups: 0 = ms_render picks render_kernel's NG from the resolution
float g_tail_rounds = -1.f;            // ms_debug_ray_group_tail: < 0 = ms_render's own share of one-group waves at the end of a launch of wide ones
int g_tail_envs = -1;                  //   ... >= 0: that many envs exactly

// -DMS_PROBE=1 (`make probe`, tools/probe_waves.py): every wave of physics_kernel and render_kernel leaves a record of
// time stamps (s_memtime at its start, at a few points where something it waited for has arrived, at its end) and of
// the SIMD it ran on, for a picture of how a launch fills and drains the chip.  Compiled out of the product library.
#ifndef MS_PROBE
#define MS_PROBE 0
#endif
#if MS_PROBE
constexpr int PROBE_STAMPS = 8;        // a record: 8 stamps (low 32 bits of s_memtime), then where the wave ran
constexpr int PROBE_WORDS = 16;        // ... the real-time counter at its start and end, and five numbers of the wave's choosing
__device__ unsigned* g_probe = nullptr;                // one record per wave, indexed by the wave's number in its launch
__device__ long long g_probe_cap = 0;
// (one VGPR: lane k holds stamp k, the low 32 bits of s_memtime - a wave's record costs the kernel one register and no
// traffic until its end; records are indexed by wave, not drawn from a cursor: thousands of atomics on one address
// would be the slowest thing in the launch)
struct Probe {
    unsigned t = 0u, real0 = (unsigned)wall_clock64();
    __device__ void done(const int lane, const long long wave) {
        if (g_probe && wave < g_probe_cap) {
            unsigned v = t;
            if (lane == PROBE_STAMPS) v = ((unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffffu) | ((unsigned)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf) << 16);   // HW_ID, XCC_ID
            if (lane == PROBE_STAMPS + 1) v = real0;                    // s_memtime counts per XCD; the 100 MHz real-time counter is the chip's
            if (lane == PROBE_STAMPS + 2) v = (unsigned)wall_clock64();
            if (lane < PROBE_WORDS) g_probe[wave*PROBE_WORDS + lane] = v;     // (lanes 11..15: numbers left by PROBE_VAL)
        }
    }
};
#define PROBE_STAMP(k) { const unsigned c_ = (unsigned)clock64(); asm volatile("v_writelane_b32 %0, %1, " #k : "+v"(probe_.t) : "s"(c_)); }
#define PROBE_INIT Probe probe_; PROBE_STAMP(0)
// a stamp once `v` (a float or an int the wave has been waiting for) is in a register
#define PROBE_AT(i, v) { asm volatile("" :: "v"(v)); PROBE_STAMP(i) }
#define PROBE_DONE(wave) { PROBE_STAMP(7) probe_.done(lane, (long long)(wave)); }
// a (wave-uniform) number instead of a time in slot k
#define PROBE_VAL(k, x) { const unsigned c_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(x)); asm volatile("v_writelane_b32 %0, %1, " #k : "+v"(probe_.t) : "s"(c_)); }
#else
#define PROBE_INIT
#define PROBE_AT(i, v)
#define PROBE_DONE(tag)
#define PROBE_VAL(k, x)
#endif

// ------------------------------------------------------------------------------------------------
// Scalar math shared by the kernels (device) and ms_host_sincospi (host).
// ------------------------------------------------------------------------------------------------

struct P2 { float x, y; };

__host__ __device__ inline P2 p2(float x, float y) { return P2{x, y}; }
__host__ __device__ inline P2 operator-(P2 a, P2 b) { return p2(a.x - b.x, a.y - b.y); }
__host__ __device__ inline P2 operator+(P2 a, P2 b) { return p2(a.x + b.x, a.y + b.y); }
__host__ __device__ inline P2 operator*(P2 a, float v) { return p2(a.x*v, a.y*v); }
__host__ __device__ inline P2 operator/(P2 a, float v) { return p2(a.x/v, a.y/v); }
__host__ __device__ inline float len2(P2 a) { return a.x*a.x + a.y*a.y; }
__host__ __device__ inline float len(P2 a) { return sqrtf(len2(a)); }
__host__ __device__ inline float cross(P2 v, P2 w) { return v.x*w.y - v.y*w.x; }
__host__ __device__ inline float dot(P2 v, P2 w) { return v.x*w.x + v.y*w.y; }

// fminf/fmaxf with NaN and signed-zero behaviour spelled out (first operand wins ties).
__host__ __device__ inline float ms_min(float a, float b) { if (a != a) return b; return (b < a) ? b : a; }
__host__ __device__ inline float ms_max(float a, float b) { if (a != a) return b; return (b > a) ? b : a; }

// direction of (x, y) in [0, 4): 0 along +x, 1 along +y, 2 along -x, 3 along -y; NaN at the origin
__host__ __device__ inline float pseudo_angle(float x, float y) {
    const float p = y/(fabsf(x) + fabsf(y));
    return x < 0.f ? 2.f - p : (p < 0.f ? 4.f + p : p);
}
// Runs of directions in 1/64ths of a pseudo-angle unit, modulo 256 (wg_arc, where the wall grid is built, has the whole
// story): the steps wa..wb (inclusive) that hold the directions from a wave's rightmost ray to its leftmost, widened by
// the same margin as the walls' arcs; and whether two such runs share a step.
constexpr float WG_ARC_MARGIN = 2e-3f;
__host__ __device__ inline void wg_wedge(const float p_right, const float p_left, int& wa8, int& wb8) {
    float width = p_left - p_right;
    width = width < 0.f ? width + 4.f : width;
    const int a = (int)floorf((p_right - WG_ARC_MARGIN)*64.f), b = (int)floorf((p_right + width + WG_ARC_MARGIN)*64.f);
    wa8 = 0; wb8 = 255;
    if (!(width < 2.f) || b - a >= 255) return;                          // (half a turn and more: fov < 180 rules it out; NaNs)
    wa8 = a & 255; wb8 = b & 255;
}
__host__ __device__ inline bool wg_arcs_meet(const int lo8, const int hi8, const int wa8, const int wb8) {
    return (((wa8 - lo8) & 255) <= ((hi8 - lo8) & 255)) | (((lo8 - wa8) & 255) <= ((wb8 - wa8) & 255));
}

// sin(pi x), cos(pi x); stands in for sinpif/cospif (kernels.cu:305-306,336-337).  The range
// reduction is exact in binary32, the kernel is a Taylor series in binary64 rounded once.
__host__ __device__ inline void sincospi_f(float x, float& s, float& c) {
    // exact in binary32: y = x - 2 rint(x/2) in [-1, 1], then z = y - rint(2y)/2 in [-1/4, 1/4]
    const float y = x - 2.f*rintf(x*0.5f);
    const float nq = rintf(2.f*y);
    const float z = y - 0.5f*nq;
    const int q = ((int)nq) & 3;
    const double zd = (double)z;
    const double w = zd*zd;
    double ps = -2.1915353447830217e-05;
    ps = ps*w + 0.00046630280576761255; ps = ps*w + -0.0073704309457143504;
    ps = ps*w + 0.08214588661112823;    ps = ps*w + -0.5992645293207921;
    ps = ps*w + 2.5501640398773455;     ps = ps*w + -5.16771278004997;
    ps = ps*w + 3.141592653589793;
    ps = ps*zd;
    double pc = 4.303069587032947e-06;
    pc = pc*w + -0.0001046381049248457; pc = pc*w + 0.0019295743094039231;
    pc = pc*w + -0.02580689139001406;   pc = pc*w + 0.2353306303588932;
    pc = pc*w + -1.3352627688545895;    pc = pc*w + 4.0587121264167685;
    pc = pc*w + -4.934802200544679;
    pc = pc*w + 1.0;
    const float S = (float)ps, C = (float)pc;
    switch (q) {
        case 0:  s =  S; c =  C; break;
        case 1:  s =  C; c = -S; break;
        case 2:  s = -S; c = -C; break;
        default: s = -C; c =  S; break;
    }
}

// ATen `%` on floats (remainder): fmod then sign fix-up.            kernels.cu:173-175
__device__ inline float remainder_f(float a, float b) {
    float m = fmodf(a, b);
    if ((m != 0.f) && ((b < 0.f) != (m < 0.f))) m += b;
    return m;
}
__device__ inline float normalize_degrees(float a) {
    return remainder_f(remainder_f(a, 360.f) + 180.f, 360.f) - 180.f;
}

struct Isect { float s, t; };
// kernels.cu:67-89
__device__ inline Isect intersect(P2 P, P2 U, P2 Q, P2 V) {
    const float UxV = cross(U, V);
    if (fabsf(UxV) < 1.e-3f) return Isect{INFINITY, INFINITY};
    const P2 PQ = Q - P;
    return Isect{cross(PQ, V)/UxV, cross(PQ, U)/UxV};
}

struct Proj { float s, d; };
// kernels.cu:91-107
__device__ inline Proj project(P2 P, P2 U, P2 Q) {
    const float u = len(U) + 1e-6f;
    const P2 PQ = Q - P;
    return Proj{dot(PQ, U)/(u*u), fabsf(cross(PQ, U))/u};
}

// kernels.cu:109-118; never returns NaN or -0, so the folds over it are order-independent.
__device__ inline float sensibilize(float p) {
    const float q = p*.99f;
    if (!(q > 0.f)) return 0.f;
    return (q < 1.f) ? q : 1.f;
}

// kernels.cu:119-133
__device__ inline float collision_cc(P2 p0, P2 v0, P2 p1, P2 v1, float agent_radius) {
    const float r = 1.001f*2.f*agent_radius;
    float x = 1.f;
    const P2 dv = v0 - v1;
    const Proj a = project(p0, dv, p1);
    if ((0 < a.s) & (a.d < r)) {
        const float backoff = sqrtf(r*r - a.d*a.d)/len(dv);
        x = ms_min(x, sensibilize(a.s - backoff));
    }
    return x;
}

// Reach cull in front of collision_cc (exact).  me, o = (position, velocity per step) of the two agents; dv their relative
// velocity, D their distance, r = 2.002 R, k = 1 + 1e-6/|dv| (project()'s "+ 1e-6").  collision_cc leaves x = 1 unless
// d = D |sin| / k < r and s - backoff = D cos /(|dv| k^2) - sqrt(r^2 - d^2)/|dv| < 1/0.99, so unless
// D < k^2 (1.0102 |dv| + r) + k r; and for dv = 0 exactly project()'s s is 0 and its test never fires.  |dv| is bounded
// from both sides by its components; 2 %, a millimetre and the positions' rounding are added.  A NaN anywhere fails the
// cull and takes the test.  (tests/test_wallgrid.py checks "apart => the oracle's collision_cc is 1" on random pairs.)
__host__ __device__ inline bool agents_apart(const float4 me, const float4 o, const float agent_radius) {
    const float ax = fabsf(me.z - o.z), ay = fabsf(me.w - o.w);
    const float v_up = ax + ay, v_lo = fmaxf(ax, ay);
    const float r2 = 1.001f*2.002f*agent_radius;
    const float kq = 1.f + 1.0001e-6f/v_lo;
    const float reach = 1.02f*(kq*kq*(1.0102f*v_up + r2) + kq*r2) + 1e-3f
                      + 1e-4f*(fabsf(me.x) + fabsf(me.y) + fabsf(o.x) + fabsf(o.y));
    const float dx = o.x - me.x, dy = o.y - me.y;
    return (0.9998f*(dx*dx + dy*dy) > reach*reach) | ((ax == 0.f) & (ay == 0.f));
}

// Reach cull in front of collision_cs (exact).  How far a wall can be from an agent and still matter: the crossing and
// side tests need it within |v| + r of p; an endpoint test (kernels.cu:147-160) needs d = D |sin| |v|/(|v| + 1e-6) < r and
// s - backoff = D cos |v|/(|v| + 1e-6)^2 - sqrt(r^2 - d^2)/|v| below 1/0.99 (beyond that the 0.99 margin clamps x to 1),
// which bounds the endpoint's distance D by k^2 (1.0102 |v| + r) + k r with k = 1 + 1e-6/|v|.  At everyday speeds k is 1
// and that is the familiar |v| + 2 r; it is project()'s "+ 1e-6" that lets a CRAWLING agent - a momentum velocity that
// has decayed for a hundred steps - be stopped by walls metres away, and k says exactly how many (reach 1.3 m at 6e-7 m
// a step, every wall of the map below 1e-8).  2 %, a millimetre and the position's rounding are added.  (p0, v0: position
// and velocity per step.  tests/test_wallgrid.py checks "beyond => the oracle's collision_cs is 1" on random pairs.)
__host__ __device__ inline float wall_reach(const P2 p0, const P2 v0, const float agent_radius) {
    const float vl = len(v0);
    const float r1 = 1.001f*agent_radius;
    const float kq = 1.f + 1e-6f/vl;                                       // (|v| = 0: inf, unused)
    const float reach = (vl > 0.f) ? 1.02f*(kq*kq*(1.0102f*vl + r1) + kq*r1) : 2.04f*r1;
    return reach + 1e-3f + 1e-4f*(fabsf(p0.x) + fabsf(p0.y));
}
__host__ __device__ inline float reach_squared(const float reach) { return (reach == reach) ? reach*reach : INFINITY; }   // NaN positions: test everything
// ... and the wall u = (ax, ay, bx, by) against it: the squared distance from the agent tk = (x, y, ..) to the segment,
// shaved so it is a lower bound (the reciprocal may be the hardware's approximate one: a foot a few ulps off the nearest
// point is farther away, not nearer).  Walls shorter than a tenth of a millimetre are never beyond: project()'s "+ 1e-6"
// on the WALL's length stretches the side test's reach for them (kernels.cu:91-107,163-168).  NaNs are never beyond.
__host__ __device__ inline bool wall_beyond(const float4 tk, const float4 u, const float reach2) {
    const float vx = u.z - u.x, vy = u.w - u.y;
    const float pqx = u.x - tk.x, pqy = u.y - tk.y;
#if defined(__HIP_DEVICE_COMPILE__)
    float tc = -(pqx*vx + pqy*vy)*__builtin_amdgcn_rcpf(vx*vx + vy*vy);
#else
    float tc = -(pqx*vx + pqy*vy)/(vx*vx + vy*vy);
#endif
    tc = fminf(fmaxf(tc, 0.f), 1.f);
    tc = (tc == tc) ? tc : 0.f;
    const float qx = pqx + tc*vx, qy = pqy + tc*vy;
    return (0.9998f*(qx*qx + qy*qy) > reach2) & (vx*vx + vy*vy >= 1e-8f);
}

// kernels.cu:135-171
__device__ inline float collision_cs(P2 p, P2 v, P2 la, P2 lb, float agent_radius) {
    const float r = 1.001f*agent_radius;
    float x = 1.f;
    const P2 lv = lb - la;
    const float vlen = len(v);
    const float dp = project(la, lv, p).d;   // used by both the crossing and the side test

    const Isect mid = intersect(p, v, la, lv);
    if ((0 < mid.s) & (mid.s < 1) & (0 < mid.t) & (mid.t < 1)) {
        x = ms_min(x, sensibilize((1 - r/dp)*mid.s));
    }
    const Proj a = project(p, v, la);
    if ((0 < a.s) & (a.d < r)) {
        const float backoff = sqrtf(r*r - a.d*a.d)/vlen;
        x = ms_min(x, sensibilize(a.s - backoff));
    }
    const Proj b = project(p, v, lb);
    if ((0 < b.s) & (b.d < r)) {
        const float backoff = sqrtf(r*r - b.d*b.d)/vlen;
        x = ms_min(x, sensibilize(b.s - backoff));
    }
    const Proj side = project(la, lv, p + v);
    if ((0 < side.s) & (side.s < 1) & (side.d < r)) {
        const float dq = side.d;
        x = ms_min(x, sensibilize((dp - r)/(dp - dq)));
    }
    return x;
}


// v_readlane_b32 of a float: broadcast lane `l` (wave-uniform) of v through an SGPR, no LDS round trip
__device__ inline float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ inline float bits_f(uint32_t u) { return __uint_as_float(u); }

// Wave-wide inclusive scans on the VALU (DPP row shifts + the gfx9 row broadcasts), no LDS traffic.
// `ident` fills lanes whose DPP source falls off the row / the masked rows.
// (hipcc does not fold update_dpp into the consuming op, so these are spelled out: one VALU op per step, the
// two wait states a DPP read of a freshly written VGPR needs are in the string, EXEC must be full.)
#define MS_SCAN6(op) \
    "s_nop 1\n " op " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
    "s_nop 1\n " op " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n" \
    "s_nop 1\n " op " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n" \
    "s_nop 1\n " op " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n" \
    "s_nop 1\n " op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n" \
    "s_nop 1\n " op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n" \
    "s_nop 1\n"
__device__ inline int wave_scan_add(int x) {                           // lanes without a DPP source keep their value
    asm volatile(MS_SCAN6("v_add_u32_dpp") : "+v"(x));
    return x;
}
[[maybe_unused]] __device__ inline int wave_scan_max(int x) {
    asm volatile(MS_SCAN6("v_max_i32_dpp") : "+v"(x));
    return x;
}
__device__ inline float wave_max_f(float x) {                          // all-lanes max of non-negative floats
    int v = __float_as_int(x);                                         // non-negative floats order like ints
    asm volatile(MS_SCAN6("v_max_i32_dpp") : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(v, 63));
}
__device__ inline uint32_t f_bits(float f) { return __float_as_uint(f); }

// One env's rows of `lines` behind a buffer descriptor.  A chunk of 64 rows is then ONE instruction with no address
// arithmetic in front of it - `buffer_load_dwordx4` takes the lane's byte offset from a VGPR that never changes and the
// chunk's from a scalar register - and rows past the end come back as zeros (the hardware's bounds check), where a
// plain load needs its index clamped.  Built from wave-uniform values only, so the descriptor lives in SGPRs.
struct LineRows {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ LineRows(const float4* base, int n_rows)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(base), 0, n_rows*16, 0x00020000)) {}
    __device__ float4 load(int lane_bytes, int first_row) const {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_bytes, first_row*16, 0);   // (first_row: 0 from every caller)
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
    // rows first_row + lane.  (The chunk's offset rides in the lane's VGPR offset, which the hardware's bounds check
    // covers for certain; the scalar offset is added after the check on some generations.)
    __device__ float4 chunk(int lane, int first_row) const { return load((first_row + lane)*16, 0); }
    __device__ float4 row(int i) const { return load(i*16, 0); }
};

// ------------------------------------------------------------------------------------------------
// physics                                                                    kernels.cu:179-230
// ------------------------------------------------------------------------------------------------
// ONE WAVEFRONT PER ENV (workgroup = 64 threads): the step is a chain of dependent loads around very little
// arithmetic, so what matters is how many envs are in flight and how few round trips each needs.  A wave asks for
// its first wall chunks before anything else, reads the agents (lane = agent) while they travel, and keeps
// PHYS_AHEAD chunks in flight through the sweep; agents of one env read each other's start-of-step state, which
// one wave orders for free (all reads sit before the first write in program order).
//
// Reach cull (exact): all four sub-tests of collision_cs leave x = 1 for a wall farther from the agent than
// 1.02|v| + 2r - the crossing and side tests need the wall within |v| + r of p, and an endpoint that far ahead
// clamps to 1 (0.99 (a.s - backoff) >= 1).  The margin dwarfs rounding - for |v| >= 1e-3; slower agents (but not
// stationary ones) are exempt from the cull, because project()'s |v| + 1e-6 distorts their distances.  Lanes test one wall each with cheap
// arithmetic; the few (agent, wall) pairs in reach are compacted into LDS and only those pay for the ten
// divides and five square roots of the real test.  Results are folded with atomicMin on the float's bits:
// every value is in [+0, 1], where the unsigned order is the float order, so the fold is exact in any order.
constexpr int PHYS_AHEAD = 4;          // wall chunks in flight per wave (six: no faster at 300 walls, 12 % slower at 1000 - fewer waves fit)
constexpr int PHYS_FEW = 4;            // up to this many agents per env, their reach boxes ride in scalar registers
constexpr int PHYS_PAIRS = (PHYS_FEW + 1)*WAVE;   // capacity of a wave's (wall, agent) pair list: a flush's worth + one chunk's worth for PHYS_FEW agents

// MOVE = 1: the movement modules' velocity update runs first (MsMovement), on the state this wave is loading anyway
// EXTRA = 1: the environment's bookkeeping (MsStepExtras: lifespans, respawns, IMU) runs in the same launch
template <int MOVE, int EXTRA>
__global__ __launch_bounds__(WAVE) void physics_kernel(
        const MsScenery sc, const MsAgents ag, float* __restrict__ progress,
        const float agent_radius, const float fps, const MsMovement mv, const MsStepExtras ex) {
    PROBE_INIT
    extern __shared__ float4 s_dyn[];            // per agent: (p, v/fps) | reach box | reach^2 | progress bits
    __shared__ float4 s_wall[PHYS_PAIRS];        // walls near ...
    __shared__ int s_tag[PHYS_PAIRS];            // ... this agent
    const int A = sc.n_agents, AF = sc.n_agents*sc.n_model;
    const int lane = threadIdx.x, n = blockIdx.x;                        // the host launches one wave per env
    float4* s_task = s_dyn;
    float4* s_box = s_task + A;
    float* s_reach2 = reinterpret_cast<float*>(s_box + A);
    unsigned* s_prog = reinterpret_cast<unsigned*>(s_reach2 + A);
    // (no __restrict__: the movement prologue and the epilogue write the same arrays through other pointers)
    const float2* pos2 = reinterpret_cast<const float2*>(ag.positions);
    const float2* vel2 = reinterpret_cast<const float2*>(ag.velocity);

    const int L = sc.lines_widths[n];
    const float4* __restrict__ ln = reinterpret_cast<const float4*>(sc.lines_vals) + sc.lines_starts[n];
    // Without a wall grid the first wall chunks are requested before anything else: nothing below depends on them until
    // the sweep, and on large maps the stream of walls is what that path lasts (unconditional buffer loads: lanes past the
    // last wall read zeros and are masked by `live` in the sweep).  With a grid the walls come from the agents' cells
    // instead, and nothing is asked for here.
    const LineRows rows(ln, L);
    const bool gridded = sc.wg_cells != nullptr;                         // (the same for every wave of the launch)
    // (the env's row of the wall grid, asked for with its other rows - where it is used, once the agents' positions are
    // known, it would be one more round trip; unconditionally: without a grid ms_step_physics points the two at rows
    // that exist)
    const float4 wg_geom_n = reinterpret_cast<const float4*>(sc.wg_geom)[n];
    const int wg_start_n = sc.wg_starts[n];
    float4 w[PHYS_AHEAD];
    #pragma unroll
    for (int k = 0; k < PHYS_AHEAD; k++) w[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!gridded) {
        #pragma unroll
        for (int k = 0; k < PHYS_AHEAD; k++) w[k] = rows.chunk(lane, AF + k*WAVE);
    }
    // one lane per agent: its state (kept for the epilogue).  (Behind a guard on purpose: asked for by every lane, the
    // last agent's re-read by the idle ones - which lets hipcc batch these loads with the env's rows - measured 8 % slower
    // at the headline shape and 7 % on 1000-wall maps.)
    float2 my_p, my_v;
    float my_w, my_ang;
    my_p = make_float2(0.f, 0.f); my_v = make_float2(0.f, 0.f); my_w = 0.f; my_ang = 0.f;
    if (lane < A) { my_p = pos2[n*A + lane]; my_v = vel2[n*A + lane]; my_w = ag.angvelocity[n*A + lane]; my_ang = ag.angles[n*A + lane]; }

    // the spawn pose of agent i, if it is to be respawned (modules.py:321-326)
    auto spawn_pose = [&](const int i, float2& p, float& ang) {
        const long long c = min(max(ex.respawn_choice[i], 0ll), (long long)ex.n_spawns - 1);
        p = reinterpret_cast<const float2*>(ex.spawn_positions)[(size_t)i*ex.n_spawns + c];
        ang = ex.spawn_angles[(size_t)i*ex.n_spawns + c];
    };
    if constexpr (EXTRA == 1) {
        for (int t = lane; t < A; t += WAVE) {
            const int i = n*A + t;
            bool reset = ex.respawn_mask && ex.respawn_mask[i];
            if (ex.lifespans) {                                          // modules.py:361-366
                int life = ex.lifespans[i] + 1;
                reset = reset | (life >= ex.max_lifespans[i]);
                if (reset) { life = 0; ex.max_lifespans[i] = ex.fresh_max[i]; }
                ex.lifespans[i] = life;
                if (ex.respawn_mask) ex.respawn_mask[i] = reset ? 1 : 0;
            }
            if (reset && ex.spawn_positions && !ex.respawn_after) {
                float2 p; float ang;
                spawn_pose(i, p, ang);
                if (t == lane) { my_p = p; my_ang = ang; my_v = make_float2(0.f, 0.f); my_w = 0.f; }
                // through memory as well: agents beyond the first 64 live there, and the movement prologue and the
                // epilogue's "velocity only changes on a collision" rule read it back
                reinterpret_cast<float2*>(ag.positions)[i] = p;
                ag.angles[i] = ang;
                reinterpret_cast<float2*>(ag.velocity)[i] = make_float2(0.f, 0.f);
                ag.angvelocity[i] = 0.f;
            }
        }
    }
    if constexpr (MOVE == 1) {
        // modules.py:57-66,106-118: look the action up, turn its velocity delta into the global frame, blend
        // (the table - seven actions, three floats each - rides in the lanes of one register, asked for up front: looked
        // up in memory by the action it would be a round trip behind the actions' own)
        const bool small_table = 3*mv.n_actions <= WAVE;
        const float tab = mv.table[min(lane, 3*mv.n_actions - 1)];
        auto moved = [&](const int i, const float ang, float2& v, float& w, const float dx, const float dy, const float dw) {
            const float a_ = 0.017453292519943295f*ang;                 // np.pi/180*angles, in binary32 like torch
            const float s_ = sinf(a_), c_ = cosf(a_);
            const float gx = c_*dx - s_*dy, gy = s_*dx + c_*dy;
            if (mv.keep == 0.f) { w = dw; v = make_float2(gx, gy); }
            else { w = mv.keep*w + dw; v = make_float2(mv.keep*v.x + gx, mv.keep*v.y + gy); }
            ag.angvelocity[i] = w;
            reinterpret_cast<float2*>(ag.velocity)[i] = v;
        };
        {
            // (every lane looks an action up - its agent's, or the last agent's again: the lanes exchange table entries, which
            // only works among lanes that are all there)
            const long long act = min(max(mv.actions[n*A + min(lane, A - 1)], 0ll), (long long)mv.n_actions - 1);
            float dx, dy, dw;
            if (small_table) { dx = __shfl(tab, 3*(int)act, WAVE); dy = __shfl(tab, 3*(int)act + 1, WAVE); dw = __shfl(tab, 3*(int)act + 2, WAVE); }
            else { dx = mv.table[3*act]; dy = mv.table[3*act + 1]; dw = mv.table[3*act + 2]; }
            if (lane < A) moved(n*A + lane, my_ang, my_v, my_w, dx, dy, dw);
        }
        for (int t = lane + WAVE; t < A; t += WAVE) {                   // agents beyond the first 64: through memory
            float2 v = vel2[n*A + t];
            float w = ag.angvelocity[n*A + t];
            const long long act = min(max(mv.actions[n*A + t], 0ll), (long long)mv.n_actions - 1);
            moved(n*A + t, ag.angles[n*A + t], v, w, mv.table[3*act], mv.table[3*act + 1], mv.table[3*act + 2]);
        }
    }
    float4 my_box = make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);   // (no agent: a box no finite wall touches)
    float my_reach = 0.f;
    for (int t = lane; t < A; t += WAVE) {
        const float2 pp = (t == lane) ? my_p : pos2[n*A + t], mm = (t == lane) ? my_v : vel2[n*A + t];
        const P2 p0 = p2(pp.x, pp.y);
        const P2 v0 = p2(mm.x, mm.y)/fps;
        const float reach = wall_reach(p0, v0, agent_radius);
        s_reach2[t] = reach_squared(reach);
        const float4 box = make_float4(p0.x - reach, p0.y - reach, p0.x + reach, p0.y + reach);   // NaNs: never rejects
        if (t == lane) { my_box = box; my_reach = (reach == reach) ? reach : INFINITY; }
        s_box[t] = box;
        s_task[t] = make_float4(p0.x, p0.y, v0.x, v0.y);
        s_prog[t] = f_bits(1.f);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    PROBE_AT(1, my_box.x)                                                // agent state has arrived
    // ... and the agent-agent tests (kernels.cu:193-200), one ordered pair per lane
    // (behind agents_apart(): agents of one env are mostly rooms apart, and then no lane of the wave goes into the test at
    // all - a fifth of a physics wave's instructions)
    for (int i = lane; i < A*A; i += WAVE) {
        const int t = i / A, d1 = i - t*A;
        if (d1 != t) {
            const float4 me = s_task[t], o = s_task[d1];
            if (!agents_apart(me, o, agent_radius)) {
                const float x = collision_cc(p2(me.x, me.y), p2(me.z, me.w), p2(o.x, o.y), p2(o.z, o.w), agent_radius);
                if (x < 1.f) atomicMin(&s_prog[t], f_bits(x));
            }
        }
    }

    // one (wall, agent) pair: the reach cull on the true distance, then the reference's test (kernels.cu:135-171,202-205)
    auto meet = [&](const float4 u, const int t) {
        const float4 tk = s_task[t];
        if (!wall_beyond(tk, u, s_reach2[t])) {
            const float x = collision_cs(p2(tk.x, tk.y), p2(tk.z, tk.w), p2(u.x, u.y), p2(u.z, u.w), agent_radius);
            if (x < 1.f) atomicMin(&s_prog[t], f_bits(x));
        }
    };
    // (wall, agent) pairs collect in an LDS list with room for one agent's worth of a chunk on top of a flush's worth.
    int cnt = 0;
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        #pragma unroll 1
        for (int p0 = 0; p0 < cnt; p0 += WAVE) {
            if (p0 + lane < cnt) meet(s_wall[p0 + lane], s_tag[p0 + lane]);
        }
        __builtin_amdgcn_wave_barrier();
        cnt = 0;
    };
    // With a wall grid (MsScenery.wg_*, wallgrid_scan_kernel): an agent's cell names every wall within wg_reach of it,
    // which is every wall the agent can touch if its own reach is no longer than that - a dozen or two instead of the
    // env's hundreds.  Lane = agent for the look-up; then the agents' lists are laid end to end and dealt to the lanes,
    // one (wall, agent) pair each.  If any agent of the env is outside its grid, or faster than the lists allow (or
    // crawling: see above), the env takes the sweep over all its walls below.
    bool swept = true;
    if (gridded) {
        unsigned first = 0u;
        int count = 0;
        bool ok = A <= WAVE;
        if (lane < A) {
            const float4 geom = wg_geom_n;
            const float inv_cell = __builtin_amdgcn_rcpf(sc.wg_cell);
            const float4 me = s_task[lane];
            const float fx = floorf((me.x - geom.x)*inv_cell), fy = floorf((me.y - geom.y)*inv_cell);
            const bool inside = (fx >= 0.f) & (fx < geom.z) & (fy >= 0.f) & (fy < geom.w);   // (NaNs: outside)
            const uint4 hdr = reinterpret_cast<const uint4*>(sc.wg_cells)[wg_start_n + (inside ? (int)fy*(int)geom.z + (int)fx : 0)];
            ok = (A <= WAVE) & inside & (my_reach <= sc.wg_reach);       // (a lane per agent: more than 64 of them take the sweep)
            first = hdr.z;
            count = ok ? (int)((my_reach <= sc.wg_reach_lo) ? (hdr.w & 0xffffu) : (hdr.w >> 16)) : 0;
        }
        PROBE_AT(2, count)                                                   // ... the cells' headers
        if (!__ballot(!ok)) {
            swept = false;
            const int incl = wave_scan_add(count);
            const int excl = incl - count;
            const int P = __builtin_amdgcn_readlane(incl, 63);
            PROBE_VAL(4, P)
            // Up to 64 pairs: one each, cull and test.  More (the envs a launch ends up waiting for: 83, 91 pairs among the
            // twelve slowest waves of a probe run against a mean of 25): the cull alone first, 64 pairs at a time, its survivors
            // laid end to end in LDS, so that the ten divides and five square roots of the test run once over full lanes
            // instead of once per round over the few lanes that got through (physics 9.0 -> 8.7 us at the headline shape).
            for (int p0 = 0; p0 < P; p0 += WAVE) {
                const int q = p0 + lane;
                int t = 0;
                for (int j = 0; j < A - 1; j++) t += (__builtin_amdgcn_readlane(incl, j) <= q) ? 1 : 0;   // whose list is pair q in?
                const int k = q - __shfl(excl, t, WAVE);
                const unsigned at = (unsigned)__shfl((int)first, t, WAVE) + (unsigned)k;
                if (P <= WAVE) {
                    if (q < P) meet(reinterpret_cast<const float4*>(sc.wg_near_rows)[at], t);
                } else {
                    float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
                    bool in_reach = false;
                    if (q < P) {
                        u = reinterpret_cast<const float4*>(sc.wg_near_rows)[at];
                        in_reach = !wall_beyond(s_task[t], u, s_reach2[t]);
                    }
                    const unsigned long long m = __ballot(in_reach);
                    if (cnt > PHYS_PAIRS - WAVE) flush();
                    if (in_reach) {
                        const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
                        s_wall[pos] = u;
                        s_tag[pos] = t;
                    }
                    cnt += __popcll(m);
                }
            }
            if (cnt) flush();
        } else {
            #pragma unroll
            for (int k = 0; k < PHYS_AHEAD; k++) w[k] = rows.chunk(lane, AF + k*WAVE);
        }
    }
    // lane = wall: which agents' reach boxes does its bounding box touch?  Those (wall, agent) pairs are compacted
    // into LDS and get the distance test and then the exact one, one pair per lane (kernels.cu:202-221)
    auto keep = [&](const int t, const unsigned long long m, const float4 u) {   // appends the lanes of `m` as (wall, agent t) pairs
        if (m) {
            if ((m >> lane) & 1ull) {
                const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
                s_wall[pos] = u;
                s_tag[pos] = t;
            }
            cnt += __popcll(m);
        }
    };
    // The usual case - a handful of agents - keeps their boxes in scalar registers, so a chunk's verdicts are four
    // compares per agent straight into lane masks and nothing in the sweep waits for the LDS.
    const bool few = A <= PHYS_FEW;
    float bx[PHYS_FEW][4];
    #pragma unroll
    for (int t = 0; t < PHYS_FEW; t++) {
        bx[t][0] = readlane_f(my_box.x, t); bx[t][1] = readlane_f(my_box.y, t);
        bx[t][2] = readlane_f(my_box.z, t); bx[t][3] = readlane_f(my_box.w, t);
    }
    for (int l0 = AF; swept && l0 < L; l0 += PHYS_AHEAD*WAVE) {
        #pragma unroll
        for (int k = 0; k < PHYS_AHEAD; k++) {
            const float4 u = w[k];
            w[k] = rows.chunk(lane, l0 + (k + PHYS_AHEAD)*WAVE);
            if (l0 + k*WAVE >= L) continue;                             // uniform
            const unsigned long long live = __ballot(l0 + k*WAVE + lane < L);
            const float x0 = fminf(u.x, u.z), x1 = fmaxf(u.x, u.z), y0 = fminf(u.y, u.w), y1 = fmaxf(u.y, u.w);
            // walls with a NaN or an infinity among their coordinates are kept whatever the boxes say
            // ... and so are walls too short for the reach argument (see meet())
            const unsigned long long odd = __ballot(!(fabsf(u.x) < INFINITY)) | __ballot(!(fabsf(u.y) < INFINITY))
                                         | __ballot(!(fabsf(u.z) < INFINITY)) | __ballot(!(fabsf(u.w) < INFINITY))
                                         | __ballot(!((u.z - u.x)*(u.z - u.x) + (u.w - u.y)*(u.w - u.y) >= 1e-8f));
            if (few & !odd) {
                unsigned long long in[PHYS_FEW], any = 0ull;            // all the verdicts first, one branch for the lot
                #pragma unroll
                for (int t = 0; t < PHYS_FEW; t++) {                    // (agents that do not exist: see my_box)
                    const unsigned long long out = __ballot(x1 < bx[t][0]) | __ballot(x0 > bx[t][2]) | __ballot(y1 < bx[t][1]) | __ballot(y0 > bx[t][3]);
                    in[t] = live & ~out;
                    any |= in[t];
                }
                if (any) {
                    if (cnt > PHYS_PAIRS - PHYS_FEW*WAVE) flush();
                    #pragma unroll
                    for (int t = 0; t < PHYS_FEW; t++) keep(t, in[t], u);
                }
            } else {
                for (int t = 0; t < A; t++) {
                    const float4 b = s_box[t];
                    const unsigned long long out = __ballot(x1 < b.x) | __ballot(x0 > b.z) | __ballot(y1 < b.y) | __ballot(y0 > b.w);
                    if (cnt > PHYS_PAIRS - WAVE) flush();
                    keep(t, live & (odd | ~out), u);
                }
            }
        }
    }
    if (cnt) flush();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    PROBE_VAL(5, swept ? 1 : 0)
    PROBE_VAL(6, __popcll(__ballot((lane < A) && (bits_f(s_prog[min(lane, A - 1)]) < 1.f))))
    PROBE_AT(3, s_prog[min(lane, A - 1)])                                // every wall has been met
    // epilogue, kernels.cu:224-227
    float2* pos2w = reinterpret_cast<float2*>(ag.positions);
    float2* vel2w = reinterpret_cast<float2*>(ag.velocity);
    for (int t = lane; t < A; t += WAVE) {
        const int i = n*A + t;
        const float x = bits_f(s_prog[t]);
        float2 p = my_p, v = my_v;
        float w_ = my_w, ang = my_ang;
        if (t != lane) { p = pos2w[i]; v = vel2w[i]; w_ = ag.angvelocity[i]; ang = ag.angles[i]; }
        p.x = p.x + x*v.x/fps;
        p.y = p.y + x*v.y/fps;
        float turned = normalize_degrees(ang + x*w_/fps);
        bool stopped = x < 1;
        if (stopped) { v = make_float2(0.f, 0.f); w_ = 0.f; }
        if constexpr (EXTRA == 1) {
            if (ex.spawn_positions && ex.respawn_after && ex.respawn_mask && ex.respawn_mask[i]) {
                spawn_pose(i, p, turned);
                v = make_float2(0.f, 0.f); w_ = 0.f;
                stopped = true;
            }
        }
        pos2w[i] = p;
        ag.angles[i] = turned;
        if (ag.headings) {                                   // what render_prep_kernel would compute, one launch earlier
            float hs, hc;
            sincospi_f(turned/180.f, hs, hc);
            reinterpret_cast<float4*>(ag.headings)[i] = make_float4(turned, hs, hc, 0.f);
        }
        if (stopped) {
            vel2w[i] = v;
            ag.angvelocity[i] = w_;
        }
        progress[i] = x;
        if constexpr (EXTRA == 1) {
            if (ex.imu) {                                    // modules.py:263-270, to_local_frame :24-31
                const float a_ = 0.017453292519943295f*turned;
                const float s_ = sinf(a_), c_ = cosf(a_);
                ex.imu[3*i] = w_/ex.imu_ang_scale;
                ex.imu[3*i + 1] = (c_*v.x + s_*v.y)/ex.imu_speed_scale;
                ex.imu[3*i + 2] = (-s_*v.x + c_*v.y)/ex.imu_speed_scale;
            }
        }
    }
    PROBE_DONE(n)
}

// ------------------------------------------------------------------------------------------------
// lighting                                                                    kernels.cu:238-268
// ------------------------------------------------------------------------------------------------
// `obstructed` for one (light, wall) pair without the two divides: with d' = |UxV| and the
// numerators sign-flipped by sign(UxV), 0 < n/d < 1  <=>  0 < n' < d' exactly in round-to-nearest
// (a quotient of two binary32 values can only round to 1 when it is 1), and 0 < s <=> 0 < c'.
// Only s < .999f needs the quotient itself.
__device__ inline bool light_blocked(P2 I, P2 U, float ax, float ay, float vx, float vy) {
    const P2 V = p2(vx, vy);
    const float UxV = cross(U, V);
    const float ad = fabsf(UxV);
    if (ad < 1.e-3f) return false;                       // (inf, inf): never obstructs
    const P2 PQ = p2(ax, ay) - I;
    const uint32_t sg = f_bits(UxV) & 0x80000000u;
    const float nt = bits_f(f_bits(cross(PQ, U)) ^ sg);
    const float cs = cross(PQ, V);
    const float ns = bits_f(f_bits(cs) ^ sg);
    if (!((nt > 0.f) & (nt < ad) & (ns > 0.f))) return false;
    return (cs/UxV) < .999f;
}

// ------------------------------------------------------------------------------------------------
// dynamic lighting with the light grid                                     kernels.cu:238-268,432-436
// ------------------------------------------------------------------------------------------------
// light_intensity() for the rays of one wavefront that landed on an agent, for sceneries that carry a light grid
// (MsScenery.lg_vals, filled by ms_bake) and have at most 64 lights per env.  Runs inside render_kernel, by the
// wave that cast the rays.  Each such ray looks up the cell its hit point is in: lights the grid marks LIT are
// unblocked, DARK ones blocked - exactly, see lightgrid_kernel - and usually that settles the ray (no UNKNOWN
// light, or the LIT ones already saturate the sum, see dynlight_kernel).  Only what is left - rays with UNKNOWN
// lights, those lights only - goes through the corridor sweep + exact tests.  Returns the intensity (for
// `dynamic` lanes); `s_pair` (LG_PAIRS entries) and `s_shadow` (128 words) are this wave's LDS scratch.
struct LightPair { float ax, ay, vx, vy, ix, iy; int light; int pad; };
constexpr int LG_PAIRS = 64;

struct LightScene {                   // what grid_light_intensity reads of an MsScenery (handed over by value)
    int n_agents, n_model;
    const float* lights_vals; const int* lights_widths; const int* lights_starts;
    const unsigned* lg_vals; const int* lg_starts; const float* lg_geom; float lg_cell;
    const unsigned* lg_list; const unsigned* lg_pool; const float4* lg_pool_rows;
    unsigned by_m_mul, by_m_sh1, by_m_sh2;       // exact division by n_model (RenderConsts.by_m), worked out by the host
};

// An env with more lights than the grid holds (it has no cells for such an env) is worked through group after group of
// 64 lights, in the lights' order - the loop below, which everyone else passes once: per group every wall is met through
// the corridor sweep, and the reference's running sum (kernels.cu:261-264) carries over from group to group.
__device__ inline float grid_light_intensity(
        const LightScene sc, const MsAgents& ag, const int n, const int lane, const bool dynamic, const int nearest_idx,
        const float cx_l, const float cy_l, const int L, const float4* __restrict__ ln,
        LightPair* s_pair, unsigned* s_shadow, unsigned& telemetry, [[maybe_unused]] unsigned* clk = nullptr) {
#if MS_PROBE
#define LG_CLK(k, v) { asm volatile("" :: "v"(v)); clk[k] = (unsigned)clock64(); }
#else
#define LG_CLK(k, v)
#endif
    const int A = sc.n_agents, AF = sc.n_agents*sc.n_model;
    const int n_lights = sc.lights_widths[n];
    const bool MANY = n_lights > WAVE;                                   // (uniform)
    float acc_in = AMBIENT;
    for (int first_light = 0; ; first_light += WAVE) {
    const int ni = min(WAVE, n_lights - first_light);
    const float* __restrict__ lights = sc.lights_vals + 3*((size_t)sc.lights_starts[n] + first_light);
    const float4 geom = reinterpret_cast<const float4*>(sc.lg_geom)[n];
    float Ix = 0.f, Iy = 0.f, Ii = 0.f;          // lane i holds light i
    if (lane < ni) { Ix = lights[3*lane]; Iy = lights[3*lane + 1]; Ii = lights[3*lane + 2]; }
    // (the agent a ray landed on: line / lines per agent, by the host's multiply-high constants - as a division by a kernel
    // argument it is two dozen instructions and three registers of reciprocal that hipcc then holds across every loop)
    int my_target = -1;
    if (dynamic) {
        const unsigned t_ = __umulhi(sc.by_m_mul, (unsigned)nearest_idx);
        my_target = (int)((t_ + (((unsigned)nearest_idx - t_) >> sc.by_m_sh1)) >> sc.by_m_sh2);
    }

    // ---- the grid's verdicts for this ray's cell (all zero = all unknown outside the grid)
    uint4 st = make_uint4(0u, 0u, 0u, 0u);
    uint2 lst = make_uint2(0u, 0u);              // the cell's candidate list: first pool word, 0x80000000 | count
    if (!MANY) {
        // (every lane reads a cell that exists - its own, or the env's first: loads without a guard overlap)
        const float fx = floorf((cx_l - geom.x)/sc.lg_cell), fy = floorf((cy_l - geom.y)/sc.lg_cell);
        const bool inside = dynamic & (fx >= 0.f) & (fx < geom.z) & (fy >= 0.f) & (fy < geom.w);
        const size_t cell_id = (size_t)sc.lg_starts[n] + (inside ? (int)fy*(int)geom.z + (int)fx : 0);
        const uint4 st_ = reinterpret_cast<const uint4*>(sc.lg_vals)[cell_id];
        uint2 lst_ = make_uint2(0u, 0u);
        if (sc.lg_list) lst_ = reinterpret_cast<const uint2*>(sc.lg_list)[cell_id];   // (uniform)
        // (component by component: a whole-vector select made hipcc keep `st` in scratch and index it)
        st.x = inside ? st_.x : 0u; st.y = inside ? st_.y : 0u; st.z = inside ? st_.z : 0u; st.w = inside ? st_.w : 0u;
        lst.x = inside ? lst_.x : 0u; lst.y = inside ? lst_.y : 0u;
    }
    LG_CLK(0, st.x + __float_as_uint(Ii))                                // the lights' rows and the cell's verdicts have arrived
#if defined(MS_LIGHT_ABLATE) && MS_LIGHT_ABLATE == 2
    return __uint_as_float(st.x ^ lst.y) + Ii;                           // (ablation: the loads and the cell look-up only)
#endif
    const bool shortcut = !MANY && __ballot((lane < ni) & !(Ii >= 0.f)) == 0ull;   // every contribution non-negative, finite
    // ---- the sum over the lights the grid proves unblocked, in light order.  Rays around one target mostly share
    // a cell, so: one pass per distinct verdict word set, scalar loop over its LIT bits (01 in the 2-bit fields)
    float part = AMBIENT;
    for (unsigned long long rem = MANY ? 0ull : __ballot(dynamic); rem; ) {
        const int j = __ffsll((long long)rem) - 1;
        const unsigned sw[4] = {(unsigned)__builtin_amdgcn_readlane((int)st.x, j), (unsigned)__builtin_amdgcn_readlane((int)st.y, j),
                                (unsigned)__builtin_amdgcn_readlane((int)st.z, j), (unsigned)__builtin_amdgcn_readlane((int)st.w, j)};
        const bool same = dynamic & (st.x == sw[0]) & (st.y == sw[1]) & (st.z == sw[2]) & (st.w == sw[3]);
        rem &= ~__ballot(same);
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            for (unsigned lw = sw[k] & ~(sw[k] >> 1) & 0x55555555u; lw; lw &= lw - 1) {
                const int i = 16*k + ((__ffs((int)lw) - 1) >> 1);
                const float d2 = len2(p2(readlane_f(Ix, i), readlane_f(Iy, i)) - p2(cx_l, cy_l));
                if (same) part += LUMINANCE*readlane_f(Ii, i)/ms_max(d2, 1.f);
            }
        }
    }
    // does the grid leave any of this ray's lights open?  (fields 00, among the first ni)
    bool has_unk = false;
    {
        const unsigned wd[4] = {st.x, st.y, st.z, st.w};
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            const int nv = min(max(ni - 16*k, 0), 16);
            const unsigned valid = (nv == 16) ? 0x55555555u : (((1u << (2*nv)) - 1u) & 0x55555555u);
            has_unk |= (~(wd[k] | (wd[k] >> 1)) & valid) != 0u;
        }
    }
    LG_CLK(1, part)                                                      // ... the sum over the LIT lights is done
#if defined(MS_LIGHT_ABLATE) && MS_LIGHT_ABLATE == 3
    return part;                                                         // (ablation: up to the sum over the LIT lights)
#endif
    // saturated: the reference's min(sum, 1) is exactly 1 whatever the unknown lights do (see dynlight_kernel)
    const bool saturated = dynamic & shortcut & (part >= 1.001f);
    const bool need = dynamic & !saturated & has_unk;
    // Everyone else is done: with no light left open the reference's in-order sum over the unblocked lights IS `part`
    if (!__ballot(need)) return MANY ? acc_in : (saturated ? 1.f : ms_min(part, 1.f));
#if defined(MS_LIGHT_ABLATE) && MS_LIGHT_ABLATE == 4
    return part;                                                         // (ablation: nothing done about open lights)
#endif
    // (`telemetry`, for the probe build only: rays with open lights, of them without a list, lists, rounds of pairs, lights)
    telemetry = 0x80000000u | (unsigned)__popcll(__ballot(need)) | ((unsigned)min(ni, 63) << 25);

    // ---- the rest is the rare path: rays with lights the grid leaves open
    auto status = [&](int i) {                   // light i's 2-bit verdict for this ray's cell; i is wave-uniform
        const unsigned wd = (i < 16) ? st.x : (i < 32) ? st.y : (i < 48) ? st.z : st.w;
        return (wd >> (2*(i & 15))) & 3u;
    };
    unsigned long long shadow = 0ull;            // open lights the walls turn out to block
    // (1) rays whose cell has a candidate list: only those (light, wall) pairs can matter anywhere in the cell.
    // Rays on one target mostly share a cell, hence a list: per distinct list its candidates are fetched once, lane =
    // candidate (entry, then the wall's row: the only dependent loads, whatever the number of rays), and left in LDS
    // with their light's position; the list's (ray, candidate) pairs are then laid end to end and dealt to the lanes,
    // 64 at a time, which read their candidate from LDS.  (Before: every pair fetched entry and wall itself, two
    // dependent round trips to cold lines per 64 pairs.)
    const bool sweep = need & (lst.y == 0u);     // no list (outside the grid, pool exhausted, ...): all the walls
    const int n_cd = (need & !sweep) ? (int)(lst.y & 0x7fffffffu) : 0;
    telemetry |= (unsigned)__popcll(__ballot(sweep)) << 7;
    if (__ballot(n_cd > 0)) {
        s_shadow[2*lane] = 0u; s_shadow[2*lane + 1] = 0u;
        for (unsigned long long lists = __ballot(n_cd > 0); lists; ) {
            const int j0 = __ffsll((long long)lists) - 1;
            const unsigned first = (unsigned)__builtin_amdgcn_readlane((int)lst.x, j0);
            const int c = __builtin_amdgcn_readlane(n_cd, j0);
            const unsigned long long members = __ballot((n_cd > 0) & (lst.x == first));
            lists &= ~members;
            telemetry += 1u << 14;
            for (int c0 = 0; c0 < c; c0 += LG_PAIRS) {
                const int nc = min(LG_PAIRS, c - c0);
                {
                    const unsigned at = first + (unsigned)(c0 + min(lane, nc - 1));
                    const unsigned e = sc.lg_pool[at];
                    const int i = (int)((e >> 24) & 63u);
                    // the candidate's wall as (a, b - a): from the pool's own copy, which arrives with the entry - or, for a
                    // scenery baked without one, from the env's lines, a trip later
                    float4 w;
                    if (sc.lg_pool_rows) w = sc.lg_pool_rows[at];            // (uniform)
                    else { const float4 u = ln[AF + (int)(e & 0xffffffu)]; w = make_float4(u.x, u.y, u.z - u.x, u.w - u.y); }
                    const float ix = __shfl(Ix, i, WAVE), iy = __shfl(Iy, i, WAVE);
                    __builtin_amdgcn_wave_barrier();                     // (the last batch's readers are through)
                    if (lane < nc) s_pair[lane] = LightPair{w.x, w.y, w.z, w.w, ix, iy, i, 0};
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                int pj = -1, pk = 0, fill = 0;       // this lane's pair: ray, candidate of the batch; lanes dealt so far
                auto round = [&]() {
                    const int src = max(pj, 0);
                    const P2 C = p2(__shfl(cx_l, src, WAVE), __shfl(cy_l, src, WAVE));
                    const LightPair pr = s_pair[pk];
                    const P2 I = p2(pr.ix, pr.iy);
                    if ((pj >= 0) && light_blocked(I, C - I, pr.ax, pr.ay, pr.vx, pr.vy))
                        atomicOr(&s_shadow[2*pj + (pr.light >> 5)], 1u << (pr.light & 31));
                    pj = -1; pk = 0; fill = 0;
                    telemetry += 1u << 18;
                };
                for (unsigned long long rays = members; rays; rays &= rays - 1) {
                    const int j = __ffsll((long long)rays) - 1;
                    for (int k0 = 0; k0 < nc; ) {
                        const int take = min(nc - k0, WAVE - fill);
                        if ((lane >= fill) & (lane < fill + take)) { pj = j; pk = k0 + lane - fill; }
                        fill += take; k0 += take;
                        if (fill == WAVE) round();
                    }
                }
                if (fill) round();
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (n_cd > 0) shadow = ((unsigned long long)s_shadow[2*lane + 1] << 32) | s_shadow[2*lane];
        __builtin_amdgcn_wave_barrier();
    }
    // (2) rays without a list: the corridor sweep over all the walls, one target agent at a time
    if (__ballot(sweep)) {
        unsigned long long need_lights = 0ull;
        for (int i = 0; i < ni; i++) if (__ballot(sweep & (status(i) == 0u))) need_lights |= 1ull << i;
        s_shadow[2*lane] = 0u; s_shadow[2*lane + 1] = 0u;
        unsigned long long todo = __ballot(sweep);
        while (todo) {
            const int target = __builtin_amdgcn_readlane(my_target, __ffsll((long long)todo) - 1);
            const bool mine = sweep & (my_target == target);
            const unsigned long long open = __ballot(mine);
            todo &= ~open;
            // the lights any of this target's rays still needs
            unsigned long long tl_mask = 0ull;
            for (unsigned long long m = need_lights; m; m &= m - 1) {
                const int i = __ffsll((long long)m) - 1;
                if (__ballot(mine & (status(i) == 0u))) tl_mask |= 1ull << i;
            }
            const float2 T = reinterpret_cast<const float2*>(ag.positions)[n*A + target];
            // extent of the hit points around the target, + float slack
            float rho = mine ? sqrtf((cx_l - T.x)*(cx_l - T.x) + (cy_l - T.y)*(cy_l - T.y)) : 0.f;
            rho = wave_max_f(rho) + 2e-3f + 1e-4f*(fabsf(T.x) + fabsf(T.y));
            // corridor frame of light `lane`: unit vector e from the light to the target, length el
            const float dx = T.x - Ix, dy = T.y - Iy;
            const float el = sqrtf(dx*dx + dy*dy);
            const float ex = dx/el, ey = dy/el;

            int cnt = 0;
            auto flush = [&]() {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const LightPair pr = s_pair[min(lane, cnt - 1)];
                const P2 I = p2(pr.ix, pr.iy);
                for (unsigned long long rays = open; rays; rays &= rays - 1) {
                    const int jr = __ffsll((long long)rays) - 1;
                    const P2 C = p2(readlane_f(cx_l, jr), readlane_f(cy_l, jr));
                    if ((lane < cnt) && light_blocked(I, C - I, pr.ax, pr.ay, pr.vx, pr.vy))
                        atomicOr(&s_shadow[2*jr + (pr.light >> 5)], 1u << (pr.light & 31));
                }
                __builtin_amdgcn_wave_barrier();
                cnt = 0;
            };
            for (int l0 = AF; l0 < L; l0 += WAVE) {
                // lane = wall: a wall can only shadow the target from a light if it reaches into the corridor
                // light -> target; surviving (wall, light) pairs go to the LDS pair list
                const bool live = l0 + lane < L;
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live) w = ln[l0 + lane];
                const float ax = w.x - T.x, ay = w.y - T.y, bx = w.z - T.x, by = w.w - T.y;
                const float m = rho + 1e-4f*(fabsf(ax) + fabsf(ay) + fabsf(bx) + fabsf(by));
                for (unsigned long long lm = tl_mask; lm; lm &= lm - 1) {
                    const int i = __ffsll((long long)lm) - 1;
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
                        if (cnt + nk > LG_PAIRS) flush();
                        if (keep) s_pair[cnt + __popcll(km & ((1ull << lane) - 1ull))] =
                            LightPair{w.x, w.y, w.z - w.x, w.w - w.y, readlane_f(Ix, i), readlane_f(Iy, i), i, 0};
                        cnt += nk;
                    }
                }
            }
            if (cnt) flush();
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (sweep) shadow = ((unsigned long long)s_shadow[2*lane + 1] << 32) | s_shadow[2*lane];
        __builtin_amdgcn_wave_barrier();
    }
    // (3) the reference's sum (kernels.cu:261-267) in light order: the grid's verdict where it has one, else the walls'.
    // Only over the lights that some open ray's cell does not call DARK (86 % of verdicts are): one pass per distinct
    // verdict word set collects them, as for `part` above.  (The sum used to visit every light, a divide each: with
    // two or three open rays in a wave and sixteen lights it was most of what the launch's last waves were doing.)
    unsigned cand[4] = {0u, 0u, 0u, 0u};         // (uniform) low bit of field i set: light i is LIT or UNKNOWN for an open ray
    for (unsigned long long rem = __ballot(need); rem; ) {
        const int j = __ffsll((long long)rem) - 1;
        const unsigned sw[4] = {(unsigned)__builtin_amdgcn_readlane((int)st.x, j), (unsigned)__builtin_amdgcn_readlane((int)st.y, j),
                                (unsigned)__builtin_amdgcn_readlane((int)st.z, j), (unsigned)__builtin_amdgcn_readlane((int)st.w, j)};
        rem &= ~__ballot(need & (st.x == sw[0]) & (st.y == sw[1]) & (st.z == sw[2]) & (st.w == sw[3]));
        #pragma unroll
        for (int k = 0; k < 4; k++) cand[k] |= ~(sw[k] >> 1) & 0x55555555u;
    }
    float acc = MANY ? acc_in : AMBIENT;
    #pragma unroll
    for (int k = 0; k < 4; k++) {
        for (unsigned lw = cand[k]; lw; lw &= lw - 1) {
            const int i = 16*k + ((__ffs((int)lw) - 1) >> 1);
            if (i >= ni) break;
            const unsigned s2 = status(i);
            const bool unblocked = (s2 == 1u) | ((s2 == 0u) & !((shadow >> i) & 1ull));
            const P2 I = p2(readlane_f(Ix, i), readlane_f(Iy, i));
            const float d2 = len2(I - p2(cx_l, cy_l));
            if (need & unblocked) acc += LUMINANCE*readlane_f(Ii, i)/ms_max(d2, 1.f);
        }
    }
    if (MANY) {
        if (first_light + WAVE >= n_lights) return ms_min(acc, 1.f);
        acc_in = acc;
        __builtin_amdgcn_wave_barrier();
        continue;
    }
    const float intensity = saturated ? 1.f : ms_min(need ? acc : part, 1.f);
    return intensity;
    }
}

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
    f.lw = rd/(ld + rd);
    f.rw = ld/(ld + rd);
    return f;
}

// First launch of ms_render when a workspace is given: zeroes the queue counter and evaluates every agent's
// sin/cos (binary64 inside, see sincospi_f) once, instead of once per wavefront of the raycast.
// Workspace layout: [0] queue length, [1] rays that took the sequential fold, [2] wavefronts that took its lane-parallel
// form (telemetry for tests) | [16, 16 + n_fans) queued ray groups | (8-byte aligned) (sin, cos) per (env, agent).
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

struct Divisor { unsigned mul, sh1, sh2; };
struct RenderConsts {
    float x_clip, c_b;
    Divisor by_f, by_g, by_m;
    Divisor by_f1, by_g1;          // render_kernel's NG > 1: the waves of one ray group at the end of every XCD's blocks (see there),
    int envs_lo, envs_rem, tail;   //   an XCD's envs (n_envs/8, the first n_envs % 8 XCDs one more) and how many of them those waves take
    int skip_own;                  // the agent's own model lines lie inside its near plane: no ray of its can hit them
    float inv_res;                 // 1/res where that is a power of two (x/res is then x*inv_res bit for bit), else 0
    int telemetry;                 // ms_debug_pair_telemetry: pair / window counts into workspace[3], [4]
};
__host__ inline Divisor divisor_of(unsigned d) {           // d >= 1
    unsigned s = 0;
    while ((1ull << s) < d) s++;
    const unsigned long long m = ((1ull << 32)*((1ull << s) - d))/d + 1ull;
    return Divisor{(unsigned)m, s < 1u ? s : 1u, s > 1u ? s - 1u : 0u};
}
__device__ inline int div_by(int n, const Divisor d) {     // n >= 0
    const unsigned t = __umulhi(d.mul, (unsigned)n);
    return (int)((t + (((unsigned)n - t) >> d.sh1)) >> d.sh2);
}

__global__ __launch_bounds__(WG) void render_prep_kernel(const MsAgents ag, int* __restrict__ workspace,
                                                         const int n_agents_total, const int n_fans) {
    const int i = blockIdx.x*WG + threadIdx.x;
    if (i == 0) { workspace[0] = 0; workspace[1] = 0; workspace[2] = 0; workspace[3] = 0; workspace[4] = 0; }
    if (i < n_agents_total) {
        float s, c;
        sincospi_f(ag.angles[i]/180.f, s, c);
        reinterpret_cast<float2*>(workspace + 16 + ((n_fans + 1) & ~1))[i] = make_float2(s, c);
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
template <int IMPL, int RW, int OBS, int SHADE = 1, int NG = 1>
// Occupancy knobs of the render kernel (A/B builds; the defaults are the product): waves per SIMD the register allocation
// is held to, chunks of rows in flight, capacity of a wave's list of visible lines (which sizes its LDS block)
#ifndef MS_WAVES
#define MS_WAVES 6
#endif
#ifndef MS_ABLATE
#define MS_ABLATE 0                          // (instruction-count experiments: 1 stops a wave after its set-up, 2 after pass 1 with
#endif                                       //  pass 2 skipped, 3 after the raycast; the outputs are then garbage)
#ifndef MS_AHEAD
#define MS_AHEAD 3
#endif
#ifndef MS_VCAP
#define MS_VCAP 128
#endif
__global__ __launch_bounds__(RW*WAVE) __attribute__((amdgpu_waves_per_eu(MS_WAVES, MS_WAVES))) void render_kernel(
        const MsScenery sc, const MsAgents ag, const MsRender out,
        const float agent_radius, const float half_screen, const int R, const int n_fans, const RenderConsts rc) {
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
    constexpr int NR = WAVE*NG;                  // rays per wave
    // (NG > 1: the list is shared by the wave's groups and must outlive their epilogues, whose scratch - the lighting's pair
    // list and shadow words, the RGB staging - therefore sits in the per-group region behind it, O_EPI, not on top of it)
    constexpr int O_EPI = (IMPL == 2 && NG > 1) ? 24*MS_VCAP + 256 : 0;
    constexpr int LDS_PER_WAVE = IMPL != 2 ? 4864 : NG == 1 ? 24*MS_VCAP + 3072 : O_EPI + 2816 + 6*MS_VCAP;
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

    // XCD-aware block order: hardware block b lands on XCD b % 8; give each XCD a contiguous run of
    // logical blocks so the fans of one env (and its lines) stay behind one L2.
    // (With one wave per workgroup the grid is exactly the fans - ms_render launches it so: the count comes from the
    // kernel's own arguments, not from the dispatch packet, and there is no early exit - either of which is a round trip
    // of its own before the loads below may even be asked for.)
    const int nb = RW == 1 ? n_fans : (int)gridDim.x, b = blockIdx.x;
    const int q8 = nb >> 3, r8 = nb & 7, xcd = b & 7, ix = b >> 3;
    const int lb = xcd*q8 + min(xcd, r8) + ix;      // (XCDs 0..r8-1 get a block more; no branch: a branch ends the stretch of
                                                    //  code hipcc gathers the kernel-argument loads of to its top)
    const int fan = NG == 1 ? lb*RW + wave : b;       // (NG > 1: the blocks' own order, see below)
    if constexpr (RW != 1) { if (fan >= n_fans) return; }                // waves are independent: no workgroup barriers below
#ifdef MS_PARK
    // (-DMS_PARK=<shader clocks>, an experiment: every render wave sits out that long before it starts, as it would at the
    // barrier of a single-launch step whose first wave does the env's physics - what do parked waves cost a launch?)
    { const long long t0_ = clock64(); while (clock64() - t0_ < MS_PARK) __builtin_amdgcn_s_sleep(8); }
#endif
    const int A = sc.n_agents, AF = sc.n_agents*sc.n_model;
    // Which rays of which agent: NG == 1, fan = (env, agent, run of 64 rays).  NG > 1, an XCD's blocks are in two parts: waves
    // of NG groups for its envs but the last rc.tail, and behind them waves of ONE group for those.  A wave of four groups
    // lives four times as long, and a launch of a few rounds of those ends with the machine draining for most of one such
    // life; the short waves are what the slots that come free take up then (ms_render sizes the second part: about half a
    // round of the long ones' work).
    int n, a, r0, span;
    if constexpr (NG == 1) {
        const int G = (R + WAVE - 1)/WAVE, F = A*G;   // g: which run of 64 rays of the agent's this wave casts
        n = div_by(fan, rc.by_f); const int rem = fan - n*F; a = div_by(rem, rc.by_g); r0 = (rem - a*G)*WAVE; span = WAVE;
    } else {
        // (XCD x takes blocks x, x + 8, ...: it is given a contiguous run of envs - an eighth of them, the first N mod 8 XCDs one
        // more - so that an env's waves and their lines stay behind one L2, its wide waves first and then the single ones of
        // its last envs; an XCD with an env fewer than the others lets its last blocks go)
        const int e_x = rc.envs_lo + (xcd < rc.envs_rem ? 1 : 0), first_x = xcd*rc.envs_lo + min(xcd, rc.envs_rem);
        const int t_x = min(rc.tail, e_x);
        const int Gw = (R + NR - 1)/NR, w_x = (e_x - t_x)*A*Gw;
        const bool single = ix >= w_x;
        const int f = single ? ix - w_x : ix;
        const int G = single ? (R + WAVE - 1)/WAVE : Gw, F = A*G;
        const Divisor df = Divisor{single ? rc.by_f1.mul : rc.by_f.mul, single ? rc.by_f1.sh1 : rc.by_f.sh1, single ? rc.by_f1.sh2 : rc.by_f.sh2};
        const Divisor dg = Divisor{single ? rc.by_g1.mul : rc.by_g.mul, single ? rc.by_g1.sh1 : rc.by_g.sh1, single ? rc.by_g1.sh2 : rc.by_g.sh2};
        const int nn = div_by(f, df), rem = f - nn*F;
        if (nn >= (single ? t_x : e_x - t_x)) return;
        a = div_by(rem, dg);
        n = first_x + (single ? e_x - t_x : 0) + nn;
        span = single ? WAVE : NR;
        r0 = (rem - a*G)*span;
    }
    const int r = r0 + lane;                       // (this lane's ray in the wave's first group)
    const int r_last = min(r0 + span - 1, R - 1);
    [[maybe_unused]] const int n_live = r_last - r0 + 1;

    const int L = sc.lines_widths[n];
    const int base = sc.lines_starts[n];
    // (the env's row of the wall grid is asked for here, with the env's other rows: where it is used - once the agent's
    // position is known - it would be one more round trip in the chain position -> cell -> list -> walls)
    // (Unconditionally: ms_render points wg_geom / wg_starts at rows that exist when there is no grid, so that these two
    // are part of the one batch of loads and not the body of a branch with a round trip of its own.)
    const float4 wg_geom_n = reinterpret_cast<const float4*>(sc.wg_geom)[n];
    const int wg_start_n = sc.wg_starts[n];
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
            const float2 sc_ = reinterpret_cast<const float2*>(out.workspace + 16 + ((n_fans + 1) & ~1))[n*A + lane];
            ag_s = sc_.x; ag_c = sc_.y;
        } else {
            const float2 sc_ = sincospi_called(ag.angles[n*A + lane]/180.f);
            ag_s = sc_.x; ag_c = sc_.y;
        }
        ag_p = reinterpret_cast<const float2*>(ag.positions)[n*A + lane];
    }
    };
    load_agents();
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
        const float uy = rc.inv_res != 0.f ? num*rc.inv_res : num/Rf;
        rx_ = cs*1.f - sn*uy; ry_ = sn*1.f + cs*uy;
        rlen_ = ray_len(rx_, ry_);
        near_ = agent_radius/rlen_;
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
    // what depends on the winner's number alone: its row, its texel count and first texel
    // (plain scalars in and out: as a struct by value this cost every wave 32 bytes of scratch memory)
    auto winner_of = [&](const int nearest_idx, float4& hw_mem, int& tex_w, int& tstart) {
        const LateArgs late = late_args();       // (see RenderArgs: read where it is used, at the wave's end)
        // (uniform; constant-folded away in the colour instantiations)
        const bool want_texel_row = COLOUR || (OBS && late->out.seen_stamp != nullptr);
        const bool want_line = want_texel_row || late->out.locations != nullptr || late->out.dots != nullptr;
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
    const bool want_texel_row = COLOUR || (OBS && late->out.seen_stamp != nullptr);
    const bool want_line = want_texel_row || late->out.locations != nullptr || late->out.dots != nullptr;
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
            loc = (pqx*ry - pqy*rx)/d;
            const float dtop = rx*vx + ry*vy;
            const float dbot = rlen*sqrtf(vx*vx + vy*vy);
            dt = dtop/(dbot + 1.e-6f);
        }
    }
    const size_t o = ((size_t)n*A + a)*R + r;
    const float dist = nearest_s*rlen;
    {
        int* const o_indices = late->out.indices;
        float* const o_locations = late->out.locations;
        float* const o_dots = late->out.dots;
        float* const o_distances = late->out.distances;
        if (r < R) {
            if (!OBS || o_indices) MS_OUT_STORE(nearest_idx, &o_indices[o]);
            if (!OBS || o_locations) MS_OUT_STORE(loc, &o_locations[o]);
            if (!OBS || o_dots) MS_OUT_STORE(dt, &o_dots[o]);
            if (!OBS || o_distances) MS_OUT_STORE(dist, &o_distances[o]);
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
            } else if (out.workspace) {
                if (lane == 0) out.workspace[16 + atomicAdd(&out.workspace[0], 1)] = fan;
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
    if constexpr (OBS == 1) {
        if (late->out.seen_stamp) {                                // explorer.py:34-58: which texels are seen for the first time
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
        if (late->out.obs_centre) {                                // deathmatch.py:74-80: who is in the crosshair
            const int sub = late->out.obs_subsample, W = R/sub;
            const int r1 = (W/2 - 1)*sub + sub/2, r2 = (W/2)*sub + sub/2;
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
        float* const o_screen = late->out.screen;
        if (!OBS || o_screen) {
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
    // ---- pooled observations (modules.py:138-145,170-184,211-224): the mean over `sub` adjacent rays of the colour
    // and of the depth 1 - clamp((distance - agent_radius)/max_depth, 0, 1), summed pairwise across lanes
    if (OBS && ((COLOUR && late->out.obs_rgb) || late->out.obs_depth)) {
        const int sub = late->out.obs_subsample;                       // power of two, divides 64 and R (checked by the host)
        float p0 = s0, p1 = s1, p2 = s2;
        float pd = 1.f - ms_min(ms_max((dist - agent_radius)/late->out.obs_max_depth, 0.f), 1.f);
        for (int o2 = 1; o2 < sub; o2 <<= 1) {
            if constexpr (COLOUR) {
                p0 += __shfl_xor(p0, o2, WAVE); p1 += __shfl_xor(p1, o2, WAVE); p2 += __shfl_xor(p2, o2, WAVE);
            }
            pd += __shfl_xor(pd, o2, WAVE);
        }
        if (((lane & (sub - 1)) == 0) & (r < R)) {
            // (the mean: a sum over a power-of-two count - the host checks - divided by it, which only moves the exponent;
            // times the exact reciprocal is the same number for a twelfth of the instructions)
            const float inv = 1.f/(float)sub;
            const int W = R/sub, px = r/sub;
            const size_t na = (size_t)n*A + a;
            if (COLOUR && late->out.obs_rgb) {
                late->out.obs_rgb[(na*3 + 0)*W + px] = p0*inv;
                late->out.obs_rgb[(na*3 + 1)*W + px] = p1*inv;
                late->out.obs_rgb[(na*3 + 2)*W + px] = p2*inv;
            }
            if (late->out.obs_depth) late->out.obs_depth[na*W + px] = pd*inv;
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
        constexpr int V_CAP = MS_VCAP, P_CAP = 4096 < 64*MS_VCAP ? 4096 : 64*MS_VCAP;
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
                    const float sv = (cd.pqx*cd.vy - cd.pqy*cd.vx)/d;        // q.s = cross(PQ, V)/UxV
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
            const_cast<unsigned*>(sc.wg_pool + wg_first), 0, listed ? 4*n_raw : 0, 0x00020000);
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
                                sv = cpv/d;
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
                                const float sv = (cd.pqx*cd.vy - cd.pqy*cd.vx)/d;
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
                                const float sv = (cd.pqx*cd.vy - cd.pqy*cd.vx)/d;
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
                            const float sv = (cd.pqx*cd.vy - cd.pqy*cd.vx)/d;        // q.s = cross(PQ, V)/UxV
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

// ------------------------------------------------------------------------------------------------
// bake                                                                        kernels.cu:270-293
// ------------------------------------------------------------------------------------------------
constexpr int BAKE_WALLS = 2048;   // occluders staged per pass: 32 KiB of LDS

__global__ __launch_bounds__(WG) void bake_kernel(const MsScenery sc) {
    __shared__ float4 s_wall[BAKE_WALLS];        // (ax, ay, vx, vy)
    const int n = blockIdx.x, tid = threadIdx.x;
    const int AF = sc.n_agents*sc.n_model;
    const int L = sc.lines_widths[n], base = sc.lines_starts[n];
    if (L == 0) return;
    const float4* __restrict__ ln = reinterpret_cast<const float4*>(sc.lines_vals) + base;
    const int num_i = sc.lights_widths[n];
    const float* __restrict__ lights = sc.lights_vals + 3*(size_t)sc.lights_starts[n];
    const int t0 = sc.textures_starts[base];
    const int t1 = sc.textures_starts[base + L - 1] + sc.textures_widths[base + L - 1];
    const int n_walls = max(L - AF, 0);
    const bool single = n_walls <= BAKE_WALLS;

    auto stage = [&](int w0) {
        const int w1 = min(w0 + BAKE_WALLS, n_walls);
        for (int i = w0 + tid; i < w1; i += WG) {
            const float4 w = ln[AF + i];
            s_wall[i - w0] = make_float4(w.x, w.y, w.z - w.x, w.w - w.y);
        }
        return w1 - w0;
    };
    int staged = 0;
    if (single) { staged = stage(0); __syncthreads(); }

    for (int tb = t0; tb < t1; tb += WG) {       // uniform trip count: barriers inside are safe
        const int t = tb + tid;
        const bool live = t < t1;
        P2 Cp = p2(0.f, 0.f);
        if (live) {
            const int l0 = sc.textures_inverse[t];
            const float loc = ((unsigned)(t - sc.textures_starts[l0]) + .5f)/sc.textures_widths[l0];
            const float4 w = reinterpret_cast<const float4*>(sc.lines_vals)[l0];
            Cp = p2(w.x, w.y)*(1.f - loc) + p2(w.z, w.w)*loc;
        }
        float acc = AMBIENT;
        for (int i0 = 0; i0 < num_i; i0 += 64) {               // lights in groups of 64 (one mask)
            const int i1 = min(i0 + 64, num_i);
            unsigned long long blocked = 0ull;
            for (int w0 = 0; w0 < n_walls; w0 += BAKE_WALLS) {
                if (!single) { __syncthreads(); staged = stage(w0); __syncthreads(); }
                for (int i = i0; i < i1; i++) {
                    const unsigned long long bit = 1ull << (i - i0);
                    bool bl = ((blocked & bit) != 0) | !live;
                    const P2 I = p2(lights[3*i], lights[3*i + 1]);
                    const P2 U = Cp - I;
                    for (int k = 0; k < staged; k++) {
                        if (__all(bl)) break;                  // every texel of the wave is in shadow already
                        const float4 w = s_wall[k];
                        bl = bl | light_blocked(I, U, w.x, w.y, w.z, w.w);
                    }
                    if (bl) blocked |= bit;
                }
            }
            for (int i = i0; i < i1; i++) {                     // accumulate in light order
                const P2 I = p2(lights[3*i], lights[3*i + 1]);
                const float d2 = len2(I - Cp);
                if (!((blocked >> (i - i0)) & 1ull)) acc += LUMINANCE*lights[3*i + 2]/ms_max(d2, 1.f);
            }
        }
        if (live) sc.baked_vals[t] = ms_min(acc, 1.f);
    }
}

// ------------------------------------------------------------------------------------------------
// bake in two phases, for sceneries that share geometry between envs and/or are large          kernels.cu:238-293
// ------------------------------------------------------------------------------------------------
// light_intensity() of a texel is  min(1, 0.1 + sum over UNBLOCKED lights of 2 I_i / max(d_i^2, 1)).  Which lights are
// blocked depends on the walls and the light positions only; the intensities I_i are per env.  So:
//   visibility_kernel  one workgroup per (representative env, light): the env's walls staged in LDS and sorted into
//                      ANGULAR BINS around the light, every texel tested against the walls of its own bin only;
//                      one bit per (texel, light) into the scratch MsScenery.bake_vis.
//   bake_sum_kernel    one thread per texel of EVERY env: the reference's in-order sum with the env's own
//                      intensities, visibility read from its representative's bits.
// Exactness of the bins: the wall a->b obstructs the point C from the light I only if the segment I->C crosses it
// (0 < t < 1 along the wall, kernels.cu:257), i.e. only if the direction of C as seen from I lies inside the arc the
// wall subtends - the shorter one between the directions of a and b.  Directions are measured with a pseudo-angle
// (monotone in the true angle, antipodes exactly 2 apart, so "shorter arc" means a difference below 2), arcs are grown
// by 2e-3 (~10^4 roundings), and anything doubtful - a light on the wall's line or at one of its ends, NaNs - goes
// into every bin.  A texel whose own direction is undefined is tested against every wall.
constexpr int BAKE_BINS = MS_BAKE_BINS;
constexpr int BAKE_ENTRIES = 6144;           // capacity of the bins' wall lists; beyond it the pass tests every wall
constexpr float BAKE_BIN_SCALE = BAKE_BINS/4.f;

// bin of a point as seen from the light; -1: undecidable
__host__ __device__ inline int bake_point_bin(P2 I, P2 C) {
    const float dx = C.x - I.x, dy = C.y - I.y;
    const float pc = pseudo_angle(dx, dy);
    if (!(pc == pc) || !(fabsf(dx) + fabsf(dy) > 1e-2f)) return -1;    // on top of the light: directions mean nothing
    const int b = (int)(pc*BAKE_BIN_SCALE);
    return b < 0 ? 0 : (b > BAKE_BINS - 1 ? BAKE_BINS - 1 : b);
}
// the circular run of bins [first, first + count) wall a->b can shadow from light I; count = BAKE_BINS: all of them
__host__ __device__ inline void bake_wall_bins(P2 I, float ax, float ay, float bx, float by, int& first, int& count) {
    constexpr float MARGIN = 2e-3f;
    const float dax = ax - I.x, day = ay - I.y, dbx = bx - I.x, dby = by - I.y;
    const float pa = pseudo_angle(dax, day), pb = pseudo_angle(dbx, dby);
    const float lo = fminf(pa, pb), hi = fmaxf(pa, pb), gap = hi - lo;
    first = 0; count = BAKE_BINS;
    // squared distance from the light to the wall
    const float vx = bx - ax, vy = by - ay;
    float tc = -(dax*vx + day*vy)/(vx*vx + vy*vy);
    tc = fminf(fmaxf(tc, 0.f), 1.f);
    tc = (tc == tc) ? tc : 0.f;
    const float qx = dax + tc*vx, qy = day + tc*vy;
    // the wall passes within a centimetre of the light, the light is (almost) on the wall's line between its
    // ends, NaNs: every bin
    if (!(pa == pa) || !(pb == pb) || !(qx*qx + qy*qy > 1e-4f) || (fabsf(gap - 2.f) < 2e-2f)) return;
    float s, e;                                                    // the arc, possibly running through 4 = 0
    if (gap < 2.f) { s = lo - MARGIN; e = hi + MARGIN; } else { s = hi - MARGIN; e = lo + 4.f + MARGIN; }
    const int bs = (int)floorf(s*BAKE_BIN_SCALE), be = (int)floorf(e*BAKE_BIN_SCALE);
    const int c = be - bs + 1;
    if (c >= BAKE_BINS) return;
    first = ((bs % BAKE_BINS) + BAKE_BINS) % BAKE_BINS;
    count = c;
}

__global__ __launch_bounds__(WG) void visibility_kernel(const MsScenery sc, const int use_bins) {
    __shared__ float4 s_wall[BAKE_WALLS];        // (ax, ay, vx, vy)
    __shared__ unsigned short s_entry[BAKE_ENTRIES];
    __shared__ int s_off[BAKE_BINS + 1];
    __shared__ int s_cursor[BAKE_BINS];
    __shared__ int s_total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // which (env, light) is this?  Global light j belongs to the last env whose lights start at or before j and
    // that has any (a binary search over lights_starts; uniform, so it runs on the scalar unit)
    const int j = blockIdx.x;
    int lo_ = 0, hi_ = sc.n_envs - 1;
    while (lo_ < hi_) {
        const int mid = (lo_ + hi_ + 1) >> 1;
        if (sc.lights_starts[mid] <= j) lo_ = mid; else hi_ = mid - 1;
    }
    int n = lo_;
    while (n > 0 && sc.lights_widths[n] == 0) n--;                       // (envs without lights share their successor's start)
    const int i = j - sc.lights_starts[n];
    if (i < 0 || i >= sc.lights_widths[n]) return;
    if (sc.env_geom && sc.env_geom[n] != n) return;                      // a member: its representative does the work
    const int AF = sc.n_agents*sc.n_model;
    const int L = sc.lines_widths[n], base = sc.lines_starts[n];
    if (L == 0) return;
    const float4* __restrict__ ln = reinterpret_cast<const float4*>(sc.lines_vals) + base;
    const int t0 = sc.textures_starts[base];
    const int t1 = sc.textures_starts[base + L - 1] + sc.textures_widths[base + L - 1];
    const int T = t1 - t0;
    const long long TB = (T + 63) >> 6;
    const long long row = sc.bake_vis_starts[n] + (long long)i*TB;
    if (row < 0 || row + TB > sc.bake_vis_words) return;                 // (the host checks this too)
    unsigned long long* __restrict__ vis = sc.bake_vis + row;
    const float* __restrict__ light = sc.lights_vals + 3*((size_t)sc.lights_starts[n] + i);
    const P2 I = p2(light[0], light[1]);
    const int n_walls = max(L - AF, 0);

    for (int w0 = 0, pass = 0; pass == 0 || w0 < n_walls; w0 += BAKE_WALLS, pass++) {   // uniform trip count
        __syncthreads();
        const int staged = max(min(BAKE_WALLS, n_walls - w0), 0);
        if (tid < BAKE_BINS) { s_off[tid] = 0; s_cursor[tid] = 0; }
        if (tid == 0) { s_off[BAKE_BINS] = 0; s_total = 0; }
        __syncthreads();
        // stage the walls; count how many land in each bin
        for (int k = tid; k < staged; k += WG) {
            const float4 w = ln[AF + w0 + k];
            s_wall[k] = make_float4(w.x, w.y, w.z - w.x, w.w - w.y);
            int first, count;
            bake_wall_bins(I, w.x, w.y, w.z, w.w, first, count);
            atomicAdd(&s_total, count);
            for (int c = 0; c < count; c++) atomicAdd(&s_off[(first + c) & (BAKE_BINS - 1)], 1);
        }
        __syncthreads();
        const bool brute = !use_bins || s_total > BAKE_ENTRIES;         // uniform
        if (!brute) {
            if (wave == 0) {                                             // exclusive scan of the 64 counts
                const int cnt = s_off[lane];
                const int incl = wave_scan_add(cnt);
                s_off[lane] = incl - cnt;
                if (lane == 63) s_off[BAKE_BINS] = incl;
            }
            __syncthreads();
            for (int k = tid; k < staged; k += WG) {
                const float4 w = ln[AF + w0 + k];                        // (not from s_wall: a + (b - a) is not b)
                int first, count;
                bake_wall_bins(I, w.x, w.y, w.z, w.w, first, count);
                for (int c = 0; c < count; c++) {
                    const int b = (first + c) & (BAKE_BINS - 1);
                    s_entry[s_off[b] + atomicAdd(&s_cursor[b], 1)] = (unsigned short)k;
                }
            }
            __syncthreads();
        }
        // every texel of the env against the walls of its bin
        for (int tb = 0; tb < T; tb += WG) {                             // uniform
            const int tl = tb + tid;
            const bool live = tl < T;
            P2 Cp = p2(0.f, 0.f);
            if (live) {
                const int l0 = sc.textures_inverse[t0 + tl];
                const float loc = ((unsigned)(t0 + tl - sc.textures_starts[l0]) + .5f)/sc.textures_widths[l0];
                const float4 w = reinterpret_cast<const float4*>(sc.lines_vals)[l0];
                Cp = p2(w.x, w.y)*(1.f - loc) + p2(w.z, w.w)*loc;
            }
            const P2 U = Cp - I;
            const long long word = (tb >> 6) + wave;
            bool bl = !live;
            if (pass > 0 && word < TB) bl |= ((vis[word] >> lane) & 1ull) != 0ull;   // blocked by an earlier pass' walls
            int e0 = 0, e1 = staged;
            bool listed = false;
            if (!brute) {
                const int b = bake_point_bin(I, Cp);
                if (b >= 0) { listed = true; e0 = s_off[b]; e1 = s_off[b + 1]; }
            }
            for (int e = e0; ; e++) {
                const bool go = !bl & (e < e1);
                if (!__any(go)) break;
                if (go) {
                    const float4 w = s_wall[listed ? (int)s_entry[e] : e];
                    bl = light_blocked(I, U, w.x, w.y, w.z, w.w);
                }
            }
            const unsigned long long m = __ballot(bl & live);
            if (lane == 0 && word < TB) vis[word] = m;
        }
    }
}

__global__ __launch_bounds__(WG) void bake_sum_kernel(const MsScenery sc) {
    const long long t = (long long)blockIdx.x*WG + threadIdx.x;
    if (t >= sc.n_texels_total) return;
    const int l0 = sc.textures_inverse[t];
    const int n = sc.lines_inverse[l0];
    const int AF = sc.n_agents*sc.n_model;
    const int L = sc.lines_widths[n], base = sc.lines_starts[n];
    const float4* __restrict__ ln = reinterpret_cast<const float4*>(sc.lines_vals) + base;
    const int num_i = sc.lights_widths[n];
    const float* __restrict__ lights = sc.lights_vals + 3*(size_t)sc.lights_starts[n];
    const int t0 = sc.textures_starts[base];
    const int t1 = sc.textures_starts[base + L - 1] + sc.textures_widths[base + L - 1];
    const long long TB = (t1 - t0 + 63) >> 6;
    const int tl = (int)(t - t0);
    const float loc = ((unsigned)(t - sc.textures_starts[l0]) + .5f)/sc.textures_widths[l0];
    const float4 w = reinterpret_cast<const float4*>(sc.lines_vals)[l0];
    const P2 Cp = p2(w.x, w.y)*(1.f - loc) + p2(w.z, w.w)*loc;
    // The texels of the agents' own lines are never looked up by ms_render (agent hits are lit dynamically,
    // kernels.cu:434) but the reference bakes them where the agents happen to stand, so they are worked out here,
    // per env, against every wall - the agents of a group's envs need not stand in the same place.
    const bool agent_line = l0 - base < AF;
    // (the rows this env's lights take must lie inside the scratch: visibility_kernel skipped them otherwise, and
    // reading on would be reading someone else's memory - such a texel keeps the ones it was initialised with)
    const long long vis_row0 = sc.bake_vis_starts[n];
    if (!agent_line && (vis_row0 < 0 || vis_row0 + (long long)num_i*TB > sc.bake_vis_words)) return;
    const unsigned long long* __restrict__ vis = sc.bake_vis + vis_row0;
    float acc = AMBIENT;
    for (int i = 0; i < num_i; i++) {                                    // kernels.cu:261-264, in light order
        const P2 I = p2(lights[3*i], lights[3*i + 1]);
        bool bl;
        if (agent_line) {
            bl = false;
            const P2 U = Cp - I;
            for (int k = AF; (k < L) & !bl; k++) {
                const float4 o = ln[k];
                bl = light_blocked(I, U, o.x, o.y, o.z - o.x, o.w - o.y);
            }
        } else {
            bl = ((vis[(long long)i*TB + (tl >> 6)] >> (tl & 63)) & 1ull) != 0ull;
        }
        const float d2 = len2(I - Cp);
        if (!bl) acc += LUMINANCE*lights[3*i + 2]/ms_max(d2, 1.f);
    }
    sc.baked_vals[t] = ms_min(acc, 1.f);
}

// ------------------------------------------------------------------------------------------------
// light grid: which lights reach which cells                   (accelerates kernels.cu:238-268 at run time)
// ------------------------------------------------------------------------------------------------
// One thread per cell of the env's grid, the env's walls staged in LDS.  For a cell (grown by LG_SLACK so a
// hit point's rounding cannot put it outside) and a light, with the reference's obstructed() test in mind:
//   LIT   if no wall comes near the corridor light -> cell: then no segment light -> point-in-cell crosses or
//         even grazes a wall, and obstructed() is false for every wall.
//   DARK  if some single wall shadows all four corners with room to spare (|UxV| >= 1e-2, t in (d, 1-d),
//         s in (d, .999-d), d = 2e-3, ~1e3 rounding errors).  For a fixed light and wall those conditions are
//         affine inequalities in the point, so they hold on the whole cell, and obstructed() is true there.
//   else  the cell stays UNKNOWN (0) for that light and ms_render tests rays in it against the walls.
//   else  the cell stays UNKNOWN (0) for that light and ms_render tests rays in it against walls - against the
//         cell's CANDIDATES for that light, the walls the LIT test could not rule out: any other wall provably
//         blocks no point of the cell (the LIT argument, wall by wall).  lightlist_kernel, a second pass, collects
//         them: (light, wall) pairs of the cell's unknown lights, stored back to back in a pool (lg_pool) that
//         cells draw from with an atomic cursor; a cell whose list does not fit (pool exhausted, or more than
//         LG_MAX_CANDS pairs - only cells far outside the walls) gets no list and its rays meet every wall.
constexpr float LG_SLACK = 0.01f;
constexpr int LG_LIGHTS = 64;          // lights per env the grid covers
constexpr int LG_MAX_CANDS = 96;       // longest candidate list a cell may have

struct LgCell {                        // a grid cell grown by LG_SLACK
    float x0, y0, x1, y1, rho;
    P2 ctr;
};
struct LgView {                        // the cell as one light sees it
    P2 I, U0, U1, U2, U3;              // light; corners relative to it
    float ex, ey, el;                  // corridor frame: unit vector light -> cell centre, its length
};

__host__ __device__ inline LgCell lg_cell_of(const float4 geom, const float cell, const int c) {
    const int nx = (int)geom.z;
    const int ix = c % nx, iy = c / nx;
    LgCell k;
    k.x0 = geom.x + ix*cell - LG_SLACK; k.y0 = geom.y + iy*cell - LG_SLACK;
    k.x1 = k.x0 + cell + 2*LG_SLACK;    k.y1 = k.y0 + cell + 2*LG_SLACK;
    k.ctr = p2(.5f*(k.x0 + k.x1), .5f*(k.y0 + k.y1));
    k.rho = .5f*sqrtf((k.x1 - k.x0)*(k.x1 - k.x0) + (k.y1 - k.y0)*(k.y1 - k.y0)) + 5e-3f + 1e-4f*(fabsf(k.ctr.x) + fabsf(k.ctr.y));
    return k;
}

__host__ __device__ inline LgView lg_view_of(const LgCell& k, const P2 I) {
    LgView v;
    v.I = I;
    const float dx = k.ctr.x - I.x, dy = k.ctr.y - I.y;
    v.el = sqrtf(dx*dx + dy*dy);
    v.ex = dx/v.el; v.ey = dy/v.el;
    v.U0 = p2(k.x0, k.y0) - I; v.U1 = p2(k.x1, k.y0) - I; v.U2 = p2(k.x1, k.y1) - I; v.U3 = p2(k.x0, k.y1) - I;
    return v;
}

// Can wall w = (ax, ay, vx, vy) shadow any point of the cell from the light?  false only when provably not.
__host__ __device__ inline bool lg_touches(const LgCell& k, const LgView& v, const float4 w) {
    const float ax = w.x - k.ctr.x, ay = w.y - k.ctr.y, bx = ax + w.z, by = ay + w.w;
    const float m = k.rho + 1e-4f*(fabsf(ax) + fabsf(ay) + fabsf(bx) + fabsf(by));
    const float ua = v.ex*ax + v.ey*ay, va = v.ex*ay - v.ey*ax;
    const float ub = v.ex*bx + v.ey*by, vb = v.ex*by - v.ey*bx;
    const bool outside = ((ua > m) & (ub > m)) | ((ua < -v.el - m) & (ub < -v.el - m)) |
                         ((va > m) & (vb > m)) | ((va < -m) & (vb < -m));
    if (outside) return false;                   // nowhere near the corridor light -> cell; NaNs fall through to true
    // Near the corridor, but does its shadow - the wedge behind the wall as seen from the light, bounded by the
    // lines light-a, light-b and the wall itself - reach the cell at all?  Not if all four corners lie, by 5 mm,
    // beyond one of those three lines.
    const P2 V = p2(w.z, w.w), PQ = p2(w.x, w.y) - v.I, PB = PQ + V;
    constexpr float MG2 = 5e-3f*5e-3f;
    const float sb = cross(PQ, PB);                                  // which side of light-a is b on
    const float la2 = len2(PQ), lb2 = len2(PB), lv2 = len2(V);
    const float a0 = cross(PQ, v.U0), a1 = cross(PQ, v.U1), a2 = cross(PQ, v.U2), a3 = cross(PQ, v.U3);
    const float b0 = cross(PB, v.U0), b1 = cross(PB, v.U1), b2 = cross(PB, v.U2), b3 = cross(PB, v.U3);
    const float si = -cross(V, PQ);                                  // which side of the wall is the light on
    const float w0_ = cross(V, v.U0 - PQ), w1_ = cross(V, v.U1 - PQ), w2_ = cross(V, v.U2 - PQ), w3_ = cross(V, v.U3 - PQ);
    auto beyond = [](float side, float c, float l2) { return (side*c < 0.f) & (c*c > MG2*l2); };
    auto same = [](float side, float c, float l2) { return (side*c > 0.f) & (c*c > MG2*l2); };
    const bool opp_a = beyond(sb, a0, la2) & beyond(sb, a1, la2) & beyond(sb, a2, la2) & beyond(sb, a3, la2);
    const bool opp_b = beyond(-sb, b0, lb2) & beyond(-sb, b1, lb2) & beyond(-sb, b2, lb2) & beyond(-sb, b3, lb2);
    const bool front = same(si, w0_, lv2) & same(si, w1_, lv2) & same(si, w2_, lv2) & same(si, w3_, lv2);
    return !(opp_a | opp_b | front);
}

// Does wall w = (ax, ay, vx, vy) shadow the whole cell from the light - all four corners, with room to spare (|UxV| >= 1e-2,
// t in (d, 1-d), s in (d, .999-d), d = 2e-3)?  For a fixed light and wall obstructed()'s conditions are affine inequalities
// in the point, so then they hold on the whole cell.
__host__ __device__ inline bool lg_shadows(const LgView& v, const float4 w) {
    const P2 V = p2(w.z, w.w), PQ = p2(w.x, w.y) - v.I;
    const float c1 = cross(PQ, V);
    const float d0 = cross(v.U0, V), d1 = cross(v.U1, V), d2 = cross(v.U2, V), d3 = cross(v.U3, V);
    const float sg = d0 < 0.f ? -1.f : 1.f;
    const float e0 = sg*d0, e1 = sg*d1, e2 = sg*d2, e3 = sg*d3, cc = sg*c1;
    bool full = (e0 >= 1e-2f) & (e1 >= 1e-2f) & (e2 >= 1e-2f) & (e3 >= 1e-2f);
    const float n0 = sg*cross(PQ, v.U0), n1 = sg*cross(PQ, v.U1), n2 = sg*cross(PQ, v.U2), n3 = sg*cross(PQ, v.U3);
    constexpr float D = 2e-3f;
    full &= (n0 > D*e0) & (n0 < (1.f - D)*e0) & (n1 > D*e1) & (n1 < (1.f - D)*e1) &
            (n2 > D*e2) & (n2 < (1.f - D)*e2) & (n3 > D*e3) & (n3 < (1.f - D)*e3);
    full &= (cc > D*e0) & (cc < (.999f - D)*e0) & (cc > D*e1) & (cc < (.999f - D)*e1) &
            (cc > D*e2) & (cc < (.999f - D)*e2) & (cc > D*e3) & (cc < (.999f - D)*e3);
    return full;
}

__global__ __launch_bounds__(WG) void lightgrid_kernel(const MsScenery sc) {
    __shared__ float4 s_wall[BAKE_WALLS];        // (ax, ay, vx, vy)
    const int n = blockIdx.y, tid = threadIdx.x;
    const float4 geom = reinterpret_cast<const float4*>(sc.lg_geom)[n];
    const int ncell = (int)geom.z*(int)geom.w;
    if ((int)blockIdx.x*WG >= ncell) return;     // uniform: whole workgroups leave together
    if (sc.env_geom && sc.env_geom[n] != n) return;   // shares its representative's grid
    const int c = blockIdx.x*WG + tid;
    const bool live = c < ncell;
    const int AF = sc.n_agents*sc.n_model;
    const int L = sc.lines_widths[n];
    const float4* __restrict__ ln = reinterpret_cast<const float4*>(sc.lines_vals) + sc.lines_starts[n];
    const int num_i = min(sc.lights_widths[n], LG_LIGHTS);
    const float* __restrict__ lights = sc.lights_vals + 3*(size_t)sc.lights_starts[n];
    const int n_walls = max(L - AF, 0);
    const LgCell k = lg_cell_of(geom, sc.lg_cell, c);
    unsigned long long touched = 0ull, dark = 0ull;

    for (int w0 = 0; w0 < n_walls; w0 += BAKE_WALLS) {       // uniform trip count: barriers are safe
        __syncthreads();
        const int staged = min(BAKE_WALLS, n_walls - w0);
        for (int i = tid; i < staged; i += WG) {
            const float4 w = ln[AF + w0 + i];
            s_wall[i] = make_float4(w.x, w.y, w.z - w.x, w.w - w.y);
        }
        __syncthreads();
        if (!live) continue;
        for (int i = 0; i < num_i; i++) {
            const unsigned long long bit = 1ull << i;
            if (dark & bit) continue;
            const LgView v = lg_view_of(k, p2(lights[3*i], lights[3*i + 1]));
            for (int j = 0; j < staged; j++) {
                const float4 w = s_wall[j];
                if (!lg_touches(k, v, w)) continue;
                touched |= bit;
                if (lg_shadows(v, w)) { dark |= bit; break; }
            }
        }
    }
    if (live) {
        unsigned wd[4] = {0u, 0u, 0u, 0u};
        for (int i = 0; i < num_i; i++) {
            const unsigned st = ((dark >> i) & 1ull) ? 2u : (((touched >> i) & 1ull) ? 0u : 1u);
            wd[i >> 4] |= st << (2*(i & 15));
        }
        reinterpret_cast<uint4*>(sc.lg_vals)[sc.lg_starts[n] + c] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
    }
}

// Second pass: the candidate lists of the cells' UNKNOWN lights.  Same thread-per-cell layout; a cell counts its
// candidates, claims that many pool words, then walks the walls again to write them.
__global__ __launch_bounds__(WG) void lightlist_kernel(const MsScenery sc) {
    __shared__ float4 s_wall[BAKE_WALLS];        // (ax, ay, vx, vy)
    const int n = blockIdx.y, tid = threadIdx.x;
    const float4 geom = reinterpret_cast<const float4*>(sc.lg_geom)[n];
    const int ncell = (int)geom.z*(int)geom.w;
    if ((int)blockIdx.x*WG >= ncell) return;
    if (sc.env_geom && sc.env_geom[n] != n) return;
    const int c = blockIdx.x*WG + tid;
    const bool live = c < ncell;
    const int AF = sc.n_agents*sc.n_model;
    const int L = sc.lines_widths[n];
    const float4* __restrict__ ln = reinterpret_cast<const float4*>(sc.lines_vals) + sc.lines_starts[n];
    const int num_i = min(sc.lights_widths[n], LG_LIGHTS);
    const float* __restrict__ lights = sc.lights_vals + 3*(size_t)sc.lights_starts[n];
    const int n_walls = max(L - AF, 0);
    const LgCell k = lg_cell_of(geom, sc.lg_cell, c);
    const size_t cell_id = (size_t)sc.lg_starts[n] + c;
    uint4 st = make_uint4(~0u, ~0u, ~0u, ~0u);
    if (live) st = reinterpret_cast<const uint4*>(sc.lg_vals)[cell_id];
    unsigned long long unk = 0ull;
    for (int i = 0; i < num_i; i++) {
        const unsigned wd = (i < 16) ? st.x : (i < 32) ? st.y : (i < 48) ? st.z : st.w;
        if (((wd >> (2*(i & 15))) & 3u) == 0u) unk |= 1ull << i;
    }
    const bool indexable = n_walls <= (1 << 24);

    int count = 0, first = 0, written = 0;
    for (int pass = 0; pass < 2; pass++) {       // 0: count, 1: write
        if (pass == 1 && live) {
            if (indexable & (count <= LG_MAX_CANDS)) {
                if (count > 0) {
                    const unsigned at = atomicAdd(&sc.lg_pool[0], (unsigned)count);
                    if ((unsigned long long)at + count + 1ull > (unsigned long long)sc.lg_pool_size) count = -1;   // pool exhausted
                    first = 1 + (int)at;
                }
            } else {
                count = -1;
            }
        }
        for (int w0 = 0; w0 < n_walls; w0 += BAKE_WALLS) {
            __syncthreads();
            const int staged = min(BAKE_WALLS, n_walls - w0);
            for (int i = tid; i < staged; i += WG) {
                const float4 w = ln[AF + w0 + i];
                s_wall[i] = make_float4(w.x, w.y, w.z - w.x, w.w - w.y);
            }
            __syncthreads();
            if (!live || count < 0 || (pass == 0 && count > LG_MAX_CANDS)) continue;
            for (unsigned long long m = unk; m; m &= m - 1) {
                const int i = __ffsll((long long)m) - 1;
                const LgView v = lg_view_of(k, p2(lights[3*i], lights[3*i + 1]));
                for (int j = 0; j < staged; j++) {
                    if (!lg_touches(k, v, s_wall[j])) continue;
                    if (pass == 0) count++;
                    else {
                        if (sc.lg_pool_rows) reinterpret_cast<float4*>(sc.lg_pool_rows)[first + written] = s_wall[j];
                        sc.lg_pool[first + written++] = 0x80000000u | ((unsigned)i << 24) | (unsigned)(w0 + j);
                    }
                }
            }
        }
    }
    if (live) {
        // [first candidate, 0x80000000 | count]; second word 0: no list
        reinterpret_cast<uint2*>(sc.lg_list)[cell_id] = count < 0 ? make_uint2(0u, 0u) : make_uint2((unsigned)first, 0x80000000u | (unsigned)written);
    }
}

// ------------------------------------------------------------------------------------------------
// wall grid: which walls matter to an agent in which cell          (accelerates kernels.cu:203-205,352-377)
// ------------------------------------------------------------------------------------------------
// The reference's raycast and its collision test meet every line of an env.  Per floorplan and per cell of a uniform
// grid over it, wallgrid_scan_kernel works out two sets of static walls:
//
// vis: the walls that can matter to a ray cast from anywhere in the cell.  A wall W is left out when ONE other wall O
// hides all of it from all of the cell:
//   (1) the cell's four corners lie on one side of O's line, at least WG `near` away from it;
//   (2) both ends of W lie on the other side;
//   (3) each of the eight segments corner -> end of W crosses O strictly inside it.
// For a fixed corner the points behind O as seen from it form a convex set, which holds both ends of W and so all of
// W; for a fixed point of W the same goes for the cell: every segment from the cell to W crosses O.  So every ray
// from the cell that hits W has hit O first - a hit the reference registers, since O is beyond the near plane (1) and
// the ray not parallel to it ((4) below) - and, by (5), computed to be nearer than W's by more than 2e-4: twice the
// 1e-4 band of the reference's order-dependent nearest-hit rule (kernels.cu:369).  Such a W cannot change the rule's
// outcome: in line order, when the fold reaches W either O came before, and the state is below s_O + 1e-4 < s_W -
// 1e-4, so W is not taken; or O comes later, and whatever the state is by then - with W taken or without - it is at
// least s_W - 1e-4 > s_O + 1e-4, so O is taken in both histories and they are one from there on.  Walls dropped from
// a set of hits one at a time, farthest first, each while its occluder is still there: the fold over what is left
// ends where the fold over all of them does.
//   (4) |V_O| dist(corner, O's line) >= 2e-3 |corner -> end of W|: then |U x V_O| >= 2e-3 for every such ray
//       (|U| >= 1), clear of the reference's 1e-3 parallelism cut-off (kernels.cu:77);
//   (5) the ends of W are behind O's line by at least
//           WG_BAND + 4 (1.2e-7 D^2 + 2.4e-7 C D) (1/h_W + 1/h_O)
//       D: the largest corner -> end distance, C: the largest coordinate, h_W / h_O: the least distance of a corner
//       from W's / O's line (W's must have the whole cell on one side too).  The bracket bounds the rounding error
//       of a hit distance as the reference computes it (a quotient of two cross products that both cancel by a factor
//       D/h), WG_BAND is 2e-4 in units of the longest ray direction vector the grid is used with (|ru| <= 8: fields
//       of view up to MS_WALLGRID_MAX_FOV degrees) with a factor 2.5 to spare.
// Walls shorter than WG_MIN_OCCLUDER are not tried as occluders (half the walls of a floorplan are the 15 cm ends of
// wall pieces, and leaving them out changes the lists by a percent); NaNs fail every comparison: such walls stay listed
// and hide nothing.
//
// near: the walls that come within wg_reach of the cell (of its centre, + half a diagonal): all that an agent in the
// cell whose step reaches no farther can touch (physics_kernel's reach cull decides wall by wall from there) - those
// within wg_reach_lo first, which is as far as an agent at an everyday speed needs to look.
constexpr float WG_SLACK = 0.01f;          // cells are grown by this on every side: a position's rounding cannot leave them
constexpr float WG_MIN_OCCLUDER = 0.3f;
constexpr float WG_BAND = 4e-3f;
constexpr float WG_MAX_RU2 = 64.f;         // |ru|^2 = 1 + tan^2(fov/2) the vis lists are good for

struct WgCell { float x0, y0, x1, y1; };

__host__ __device__ inline WgCell wg_cell_of(const float4 geom, const float cell, const int c) {
    const int nx = (int)geom.z;
    const int ix = c % nx, iy = c / nx;
    WgCell k;
    k.x0 = geom.x + ix*cell - WG_SLACK; k.y0 = geom.y + iy*cell - WG_SLACK;
    k.x1 = k.x0 + cell + 2*WG_SLACK;    k.y1 = k.y0 + cell + 2*WG_SLACK;
    return k;
}

// What the test needs of a target wall W and the cell, worked out once per (cell, W)
struct WgTarget {
    float qx[2], qy[2];           // ends of W
    float dx[8], dy[8], dp[8];    // per (corner i, end j) at 2 i + j: D = q_j - p_i, cross(D, p_i)
    float ms[8];                  // straddle margin x |D|
    float dmax, a_w, k_w;         // D; WG_BAND + K/h_W; K = 4 (1.2e-7 D^2 + 2.4e-7 C D)
    bool cullable;
};

__host__ __device__ inline WgTarget wg_target(const WgCell& k, const float4 w) {
    WgTarget t;
    t.qx[0] = w.x; t.qy[0] = w.y; t.qx[1] = w.z; t.qy[1] = w.w;
    const float px[4] = {k.x0, k.x1, k.x1, k.x0}, py[4] = {k.y0, k.y0, k.y1, k.y1};
    float d2max = 0.f;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) {
        const float dx = t.qx[j] - px[i], dy = t.qy[j] - py[i];
        const float d2 = dx*dx + dy*dy;
        t.dx[2*i + j] = dx; t.dy[2*i + j] = dy; t.dp[2*i + j] = dx*py[i] - dy*px[i];
        const float d = sqrtf(d2);
        t.ms[2*i + j] = (0.01f + 1e-4f*d)*d;
        d2max = fmaxf(d2max, d2);
    }
    t.dmax = sqrtf(d2max);
    const float cmax = fmaxf(fmaxf(fabsf(w.x), fabsf(w.y)), fmaxf(fabsf(w.z), fabsf(w.w))) + fmaxf(fabsf(k.x0), fabsf(k.x1)) + fmaxf(fabsf(k.y0), fabsf(k.y1));
    t.k_w = 4.f*(1.2e-7f*d2max + 2.4e-7f*cmax*t.dmax);
    // the cell against W's own line: all four corners on one side, the nearest h_W away
    const float vx = w.z - w.x, vy = w.w - w.y;
    const float vl = sqrtf(vx*vx + vy*vy);
    float hmin = INFINITY;
    bool pos = true, neg = true;
    for (int i = 0; i < 4; i++) {
        const float c = vx*(py[i] - w.y) - vy*(px[i] - w.x);
        pos &= c > 0.f; neg &= c < 0.f;
        hmin = fminf(hmin, fabsf(c));
    }
    const float h_w = hmin/vl;
    t.cullable = (pos | neg) & (h_w > 0.f) & (t.dmax < INFINITY);      // (NaNs, zero-length walls, a cell on W's line: never culled)
    t.a_w = WG_BAND + t.k_w/h_w;
    return t;
}

// Does wall o = (ax, ay, bx, by) hide the target from the whole cell?  true only when (1)-(5) hold.
__host__ __device__ inline bool wg_hides(const WgCell& k, const WgTarget& t, const float4 o, const float near_plane) {
    const float ax = o.x, ay = o.y, vx = o.z - o.x, vy = o.w - o.y;
    const float vl2 = vx*vx + vy*vy;
    if (!(vl2 >= WG_MIN_OCCLUDER*WG_MIN_OCCLUDER) || !t.cullable) return false;
    const float vl = sqrtf(vl2);
    // (1) the corners: c_i = cross(V, p_i - a) = |V| x signed distance
    const float c0 = vx*(k.y0 - ay) - vy*(k.x0 - ax), c1 = vx*(k.y0 - ay) - vy*(k.x1 - ax);
    const float c2 = vx*(k.y1 - ay) - vy*(k.x1 - ax), c3 = vx*(k.y1 - ay) - vy*(k.x0 - ax);
    const bool pos = (c0 > 0.f) & (c1 > 0.f) & (c2 > 0.f) & (c3 > 0.f), neg = (c0 < 0.f) & (c1 < 0.f) & (c2 < 0.f) & (c3 < 0.f);
    const float cmin = fminf(fminf(fabsf(c0), fabsf(c1)), fminf(fabsf(c2), fabsf(c3)));
    if (!((pos | neg) & (cmin >= near_plane*vl))) return false;
    // (2) the ends of W on the other side, (5) far enough behind
    const float d0 = vx*(t.qy[0] - ay) - vy*(t.qx[0] - ax), d1 = vx*(t.qy[1] - ay) - vy*(t.qx[1] - ax);
    const bool behind = pos ? ((d0 < 0.f) & (d1 < 0.f)) : ((d0 > 0.f) & (d1 > 0.f));
    const float gap = fminf(fabsf(d0), fabsf(d1));                       // x |V|
    const float need = t.a_w + t.k_w*(vl/cmin);
    if (!(behind & (gap >= need*vl))) return false;
    // (4) the reference registers the hit on O
    if (!(cmin >= 2e-3f*t.dmax)) return false;
    // (3) a and b strictly on opposite sides of every segment corner -> end
    const float bx = o.z, by = o.w;
    bool ok = true;
    for (int e = 0; e < 8; e++) {
        const float sa = t.dx[e]*ay - t.dy[e]*ax - t.dp[e];              // cross(D, a - p)
        const float sb = t.dx[e]*by - t.dy[e]*bx - t.dp[e];
        ok &= ((sa > t.ms[e]) & (sb < -t.ms[e])) | ((sa < -t.ms[e]) & (sb > t.ms[e]));
    }
    return ok;
}

// Does wall w come within `reach` of the cell?  Distance from the cell's centre to the wall, against reach + half a
// diagonal; false only when provably not (NaNs stay in: the reference stops an agent at such a wall, kernels.cu:109-118)
__host__ __device__ inline bool wg_close(const WgCell& k, const float4 w, const float reach) {
    const float cx = .5f*(k.x0 + k.x1), cy = .5f*(k.y0 + k.y1);
    const float vx = w.z - w.x, vy = w.w - w.y, pqx = w.x - cx, pqy = w.y - cy;
    float tc = -(pqx*vx + pqy*vy)/(vx*vx + vy*vy);
    tc = fminf(fmaxf(tc, 0.f), 1.f);
    tc = (tc == tc) ? tc : 0.f;
    const float qx = pqx + tc*vx, qy = pqy + tc*vy;
    const float rr = reach + .7072f*(k.x1 - k.x0) + 1e-3f + 1e-4f*(fabsf(cx) + fabsf(cy));
    return !(0.9998f*(qx*qx + qy*qy) > rr*rr) | !(vx*vx + vy*vy >= 1e-8f);   // (walls too short for the reach argument: physics_kernel's meet())
}

// From which directions can wall w be seen from the cell?  The directions from the points of the cell to the points of
// the wall are the directions of the points of the Minkowski difference wall - cell, a convex polygon spanned by (end
// of the wall) - (corner of the cell): clear of the origin - the wall clear of the cell - they form one arc, bounded by
// two of those eight.  Measured as pseudo-angles (pseudo_angle: monotone in the angle, antipodes exactly 2 apart, a full
// turn 4), widened by WG_ARC_MARGIN and quantised outwards to 1/64ths: the arc runs from step lo to step hi inclusive,
// modulo 256.  A wall that comes near the cell, or whose arc is undefined, gets the full turn (0, 255).  A ray from the
// cell hits the wall only if its direction lies in the arc: render_kernel drops listed walls whose arc misses its rays'.
__host__ __device__ inline void wg_arc(const WgCell& k, const float4 w, int& lo8, int& hi8) {
    lo8 = 0; hi8 = 255;
    const float cx = .5f*(k.x0 + k.x1), cy = .5f*(k.y0 + k.y1);
    const float vx = w.z - w.x, vy = w.w - w.y, pqx = w.x - cx, pqy = w.y - cy;
    float tc = -(pqx*vx + pqy*vy)/(vx*vx + vy*vy);
    tc = fminf(fmaxf(tc, 0.f), 1.f);
    tc = (tc == tc) ? tc : 0.f;
    const float qx = pqx + tc*vx, qy = pqy + tc*vy;
    const float rr = .7072f*(k.x1 - k.x0) + 2e-2f + 1e-4f*(fabsf(cx) + fabsf(cy));
    if (!(0.9998f*(qx*qx + qy*qy) > rr*rr)) return;                      // within a whisker of the cell (or NaN)
    const float pr = pseudo_angle(.5f*(w.x + w.z) - cx, .5f*(w.y + w.w) - cy);   // a direction in the middle of the arc
    const float px[4] = {k.x0, k.x1, k.x1, k.x0}, py[4] = {k.y0, k.y0, k.y1, k.y1};
    float lo = INFINITY, hi = -INFINITY;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) {
        float d = pseudo_angle((j ? w.z : w.x) - px[i], (j ? w.w : w.y) - py[i]) - pr;
        d = d > 2.f ? d - 4.f : (d <= -2.f ? d + 4.f : d);
        if (!(d == d)) return;
        lo = fminf(lo, d); hi = fmaxf(hi, d);
    }
    if (!(hi - lo < 1.9f) || !(pr == pr)) return;                        // (an arc is less than half a turn)
    const float a0 = pr + lo - WG_ARC_MARGIN, a1 = pr + hi + WG_ARC_MARGIN;
    lo8 = (int)floorf(a0*64.f) & 255;
    hi8 = (int)floorf(a1*64.f) & 255;
}
// The scan, one workgroup per GROUP of cells that share their candidates:
//   * with a parent grid (coarser cells, scanned before): the cells inside one parent cell.  A wall hidden from the
//     parent cell is hidden from every cell inside it - by the same occluder - so only the parent's vis list needs looking
//     at, as targets and as occluders (an occluder that is itself hidden from the parent cell has one in front of it that
//     hides whatever it hides: one is always on the list); and a wall within reach of a cell is within reach of its
//     parent (whose half diagonal covers the distance between the centres).  Two levels cut the work of a 1000-wall
//     floorplan by an order of magnitude.
//   * without one: WG_GROUP consecutive cells, every wall a candidate.
// The group's occluders (candidates long enough to be tried, WG_STAGE at most - beyond that the rest are not tried, which
// only lengthens lists) are staged in LDS once; then lane = candidate wall, one (cell, 64 candidates) item per wave at a
// time, results OR-ed into the cells' bitmaps (one bit per wall; rows: vis, near within wg_reach_lo, near beyond that)
// and counted once the group is through.
constexpr int WG_GROUP = 4, WG_STAGE = 2048, WG_ROWS = 3;

struct WgParent { const unsigned* cells; const int* starts; const float* geom; float cell; const unsigned short* pool; };

__global__ __launch_bounds__(WG) void wallgrid_scan_kernel(const MsScenery sc, const WgParent parent, const int* __restrict__ reps,
                                                          const long long* __restrict__ bits_starts, unsigned* __restrict__ bits,
                                                          unsigned* __restrict__ counts) {
    __shared__ float4 s_occ[WG_STAGE];
    __shared__ unsigned short s_occ_id[WG_STAGE];
    __shared__ int s_n_occ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = reps[blockIdx.y];
    const float4 geom = reinterpret_cast<const float4*>(sc.wg_geom)[n];
    const int ncx = (int)geom.z, ncy = (int)geom.w, ncell = ncx*ncy;
    const int AF = sc.n_agents*sc.n_model;
    const int n_walls = max(sc.lines_widths[n] - AF, 0);
    if (ncell == 0 || n_walls == 0) return;
    const float4* __restrict__ ln = reinterpret_cast<const float4*>(sc.lines_vals) + sc.lines_starts[n] + AF;
    // the group's cells (a rectangle of gw x gh cells from (gx0, gy0), or a run of the row-major order) and candidates
    int gx0 = 0, gy0 = 0, gw = 0, n_group = 0, first_cell = 0;
    const unsigned short* cand_vis = nullptr;                            // NULL: every wall
    const unsigned short* cand_near = nullptr;
    int n_vis = n_walls, n_near = n_walls;
    if (parent.cells) {
        const float4 pgeom = reinterpret_cast<const float4*>(parent.geom)[n];
        const int pcx = (int)pgeom.z, P = blockIdx.x;
        if (P >= pcx*(int)pgeom.w) return;
        const int ratio = (int)rintf(parent.cell/sc.wg_cell);
        gx0 = (P % pcx)*ratio; gy0 = (P/pcx)*ratio;
        gw = min(ratio, ncx - gx0);
        const int gh = min(ratio, ncy - gy0);
        if (gw <= 0 || gh <= 0) return;
        n_group = gw*gh;
        const uint4 hdr = reinterpret_cast<const uint4*>(parent.cells)[(size_t)parent.starts[n] + P];
        cand_vis = parent.pool + hdr.x; n_vis = (int)hdr.y;
        cand_near = parent.pool + hdr.z; n_near = (int)(hdr.w >> 16);
    } else {
        first_cell = blockIdx.x*WG_GROUP;
        if (first_cell >= ncell) return;
        n_group = min(WG_GROUP, ncell - first_cell);
    }
    auto cell_of = [&](const int j) { return parent.cells ? (gy0 + j/gw)*ncx + gx0 + j % gw : first_cell + j; };
    // stage the occluders: candidates of WG_MIN_OCCLUDER and more (in no particular order)
    if (tid == 0) s_n_occ = 0;
    __syncthreads();
    for (int i = tid; i < n_vis; i += WG) {
        const int id = cand_vis ? (int)cand_vis[i] : i;
        const float4 w = ln[id];
        const float vx = w.z - w.x, vy = w.w - w.y;
        if (vx*vx + vy*vy >= WG_MIN_OCCLUDER*WG_MIN_OCCLUDER) {
            const int at = atomicAdd(&s_n_occ, 1);
            if (at < WG_STAGE) { s_occ[at] = w; s_occ_id[at] = (unsigned short)id; }
        }
    }
    __syncthreads();
    const int n_occ = min(s_n_occ, WG_STAGE);
    const int W32 = (n_walls + 31) >> 5;
    unsigned* __restrict__ rows = bits + bits_starts[n];
    // vis: lane = candidate, every staged occluder in turn (uniform LDS reads)
    const int vis_chunks = (n_vis + WAVE - 1)/WAVE;
    for (int item = wave; item < n_group*vis_chunks; item += WAVES) {
        const int j = item/vis_chunks, i = (item - j*vis_chunks)*WAVE + lane;
        const int c = cell_of(j);
        const bool live = i < n_vis;
        const int id = cand_vis ? (int)cand_vis[min(i, n_vis - 1)] : min(i, n_vis - 1);
        const WgCell k = wg_cell_of(geom, sc.wg_cell, c);
        const WgTarget tg = wg_target(k, ln[id]);
        bool hidden = !live;
        for (int o = 0; o < n_occ; o++) {
            if (__all(hidden)) break;
            if (((int)s_occ_id[o] != id) && wg_hides(k, tg, s_occ[o], sc.wg_near)) hidden = true;
        }
        if (!hidden) atomicOr(&rows[(long long)(WG_ROWS*c)*W32 + (id >> 5)], 1u << (id & 31));
    }
    // near: lane = candidate
    const int near_chunks = (n_near + WAVE - 1)/WAVE;
    for (int item = wave; item < n_group*near_chunks; item += WAVES) {
        const int j = item/near_chunks, i = (item - j*near_chunks)*WAVE + lane;
        const int c = cell_of(j);
        if (i < n_near) {
            const int id = cand_near ? (int)cand_near[i] : i;
            const WgCell k = wg_cell_of(geom, sc.wg_cell, c);
            const float4 w = ln[id];
            if (wg_close(k, w, sc.wg_reach)) {
                const int row = wg_close(k, w, sc.wg_reach_lo) ? 1 : 2;
                atomicOr(&rows[(long long)(WG_ROWS*c + row)*W32 + (id >> 5)], 1u << (id & 31));
            }
        }
    }
    // count the group's bitmaps (atomics read what the group's other waves left in the L2)
    __syncthreads();
    for (int i = tid; i < n_group*WG_ROWS*W32; i += WG) {
        const int j = i/(WG_ROWS*W32), r = (i - j*WG_ROWS*W32)/W32, wd = i - (j*WG_ROWS + r)*W32;
        const int c = cell_of(j);
        const unsigned m = atomicOr(&rows[(long long)(WG_ROWS*c + r)*W32 + wd], 0u);
        if (m) atomicAdd(&counts[WG_ROWS*((size_t)sc.wg_starts[n] + c) + r], (unsigned)__popc(m));
    }
}

// One wavefront per (representative env, cell, list): the set bits of the cell's rows, in order, into the pools - the vis
// list as wall indices, each with the arc of directions the wall can be seen in from the cell (wg_arc) in its upper half;
// the near list (the walls within wg_reach_lo first, then the others) as the walls' rows themselves (physics_kernel wants
// nothing else of them, and saves a round trip).  A parent level for the next scan (vis_entries and near_rows NULL) gets
// both lists as bare 16-bit indices in `pool`.
__global__ __launch_bounds__(WG) void wallgrid_fill_kernel(const MsScenery sc, const int* __restrict__ reps,
                                                          const long long* __restrict__ bits_starts, const unsigned* __restrict__ bits,
                                                          unsigned short* __restrict__ pool, float4* __restrict__ near_rows,
                                                          unsigned* __restrict__ vis_entries) {
    const int lane = threadIdx.x & 63;
    const int n = reps[blockIdx.y];
    const long long item = (long long)blockIdx.x*WAVES + (threadIdx.x >> 6);
    const int c = (int)(item >> 1), kind = (int)(item & 1);
    const float4 geom = reinterpret_cast<const float4*>(sc.wg_geom)[n];
    if (c >= (int)geom.z*(int)geom.w) return;
    const int AF = sc.n_agents*sc.n_model;
    const int n_walls = max(sc.lines_widths[n] - AF, 0);
    const float4* __restrict__ ln = reinterpret_cast<const float4*>(sc.lines_vals) + sc.lines_starts[n] + AF;
    const int W32 = (n_walls + 31) >> 5;
    const uint4 hdr = reinterpret_cast<const uint4*>(sc.wg_cells)[(size_t)sc.wg_starts[n] + c];
    unsigned at = kind ? hdr.z : hdr.x;
    for (int r = kind; r < (kind ? WG_ROWS : 1); r++) {
        const unsigned* __restrict__ row = bits + bits_starts[n] + (long long)(WG_ROWS*c + r)*W32;
        for (int w0 = 0; w0 < W32; w0 += WAVE) {                         // lane = word
            const unsigned m = (w0 + lane < W32) ? row[w0 + lane] : 0u;
            const int cnt = __popc(m);
            const int incl = wave_scan_add(cnt);
            unsigned o = at + (unsigned)(incl - cnt);
            for (unsigned rest = m; rest; rest &= rest - 1) {
                const int id = 32*(w0 + lane) + __ffs((int)rest) - 1;
                if (kind && near_rows) near_rows[o++] = ln[id];
                else if (!kind && vis_entries) {                             // wall | first step of its arc << 16 | last << 24
                    int lo8, hi8;
                    wg_arc(wg_cell_of(geom, sc.wg_cell, c), ln[id], lo8, hi8);
                    vis_entries[o++] = (unsigned)id | ((unsigned)lo8 << 16) | ((unsigned)hi8 << 24);
                } else pool[o++] = (unsigned short)id;
            }
            at += (unsigned)__builtin_amdgcn_readlane(incl, 63);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side of the C-ABI
// ------------------------------------------------------------------------------------------------
int hip_fail(hipError_t e) { g_last_hip_error = (int)e; return MS_EHIP; }

bool scenery_ok(const MsScenery* s) {
    return s && s->n_envs > 0 && s->n_agents > 0 && s->n_model > 0 && s->lines_vals && s->lines_widths &&
           s->lines_starts && s->model && ((uintptr_t)s->lines_vals % 16 == 0) && ((uintptr_t)s->model % 16 == 0);
}
bool agents_ok(const MsAgents* a) { return a && a->angles && a->positions && a->angvelocity && a->velocity; }
bool config_ok(const MsConfig* c) {
    return c && c->res > 0 && c->fps > 0.f && c->agent_radius > 0.f && c->fov > 0.f && c->fov < 180.f;
}

}  // namespace

extern "C" {

int ms_abi_version(void) { return MS_ABI_VERSION; }

const char* ms_strerror(int code) {
    switch (code) {
        case MS_OK: return "ok";
        case MS_EINVAL: return "invalid argument (null/misaligned pointer, non-positive size or bad config)";
        case MS_EHIP: return "a HIP runtime call failed (see ms_last_hip_error)";
        case MS_EUNSUPPORTED: return "shape not supported by the gfx950 kernels";
        case MS_ENODEVICE: return "no HIP device visible";
        default: return "unknown megastep_hip error";
    }
}

int ms_last_hip_error(void) { return g_last_hip_error; }

int ms_device_count(void) {
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { g_last_hip_error = (int)e; return MS_ENODEVICE; }
    return n;
}

void ms_host_sincospi(float x, float* s, float* c) { sincospi_f(x, *s, *c); }

int ms_host_bake_point_bin(float light_x, float light_y, float x, float y) { return bake_point_bin(p2(light_x, light_y), p2(x, y)); }
void ms_host_bake_wall_bins(float light_x, float light_y, float ax, float ay, float bx, float by, int* first, int* count) {
    bake_wall_bins(p2(light_x, light_y), ax, ay, bx, by, *first, *count);
}

#if MS_PROBE
// (probe builds only, not part of the ABI) buf: device memory of capacity records of 11 32-bit words (8 stamps, HW_ID |
// XCC_ID << 16, the real-time counter at the wave's start and end), or NULL to stop recording
int ms_debug_probe(unsigned* buf, long long capacity) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &buf, sizeof buf) != hipSuccess) return hip_fail(hipGetLastError());
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_probe_cap), &capacity, sizeof capacity) != hipSuccess) return hip_fail(hipGetLastError());
    return MS_OK;
}
#endif

int ms_debug_ray_groups(int groups) { g_ray_groups = groups; return MS_OK; }
int ms_debug_ray_group_tail(float rounds, int envs) { g_tail_rounds = rounds; g_tail_envs = envs; return MS_OK; }
int ms_debug_pair_telemetry(int on) { g_pair_telemetry = on ? 1 : 0; return MS_OK; }

void ms_host_ray_interval_wide(const float* pose, const float* line, int res, float fov, float agent_radius, int groups, int wave,
                               int* first, int* count) {
    // (the launch-invariant values as ms_render works them out, the per-wave ones as render_kernel does)
    const float half_screen = tanf(3.14159265358979323846f/180.f*fov/2.);
    const float x_clip = 0.5f*agent_radius/sqrtf(1.f + half_screen*half_screen), c_b = 0.5f*(float)res/half_screen;
    const int nr = WAVE*groups, r0 = wave*nr;
    const float c_a = 0.5f*((float)res - 1.f), g0 = (float)r0;
    const int r_last = (r0 + nr - 1 < res - 1) ? r0 + nr - 1 : res - 1;
    const float last_local = (float)(r_last - r0);
    float xa, ya, xb, yb;
    agent_frame(pose[3], pose[2], line[0] - pose[0], line[1] - pose[1], line[2] - pose[0], line[3] - pose[1], xa, ya, xb, yb);
    ray_interval<(MS_V2_OPTS & 2) ? 1 : 0>(xa, ya, xb, yb, true, x_clip, c_a, c_b, g0, last_local, *first, *count, (float)nr);
}
void ms_host_ray_interval(const float* pose, const float* line, int res, float fov, float agent_radius, int group, int* first, int* count) {
    ms_host_ray_interval_wide(pose, line, res, fov, agent_radius, 1, group, first, count);
}

int ms_host_lightgrid_cell(const float* walls, int n_walls, const float* lights, int n_lights, float ox, float oy, int nx, int ny,
                           float cell, int c, unsigned* words, unsigned* candidates, int max_candidates) {
    // lightgrid_kernel's verdicts and lightlist_kernel's candidates for one cell, from the predicates those are compiled from
    const LgCell k = lg_cell_of(make_float4(ox, oy, (float)nx, (float)ny), cell, c);
    const int num_i = n_lights < LG_LIGHTS ? n_lights : LG_LIGHTS;
    words[0] = words[1] = words[2] = words[3] = 0u;
    int count = 0;
    for (int i = 0; i < num_i; i++) {
        const LgView v = lg_view_of(k, p2(lights[3*i], lights[3*i + 1]));
        bool touched = false, dark = false;
        for (int j = 0; j < n_walls && !dark; j++) {
            const float4 w = make_float4(walls[4*j], walls[4*j + 1], walls[4*j + 2] - walls[4*j], walls[4*j + 3] - walls[4*j + 1]);
            if (!lg_touches(k, v, w)) continue;
            touched = true;
            dark = lg_shadows(v, w);
        }
        const unsigned st = dark ? 2u : (touched ? 0u : 1u);
        words[i >> 4] |= st << (2*(i & 15));
        if (st == 0u) {
            for (int j = 0; j < n_walls; j++) {
                const float4 w = make_float4(walls[4*j], walls[4*j + 1], walls[4*j + 2] - walls[4*j], walls[4*j + 3] - walls[4*j + 1]);
                if (!lg_touches(k, v, w)) continue;
                if (count < max_candidates) candidates[count] = 0x80000000u | ((unsigned)i << 24) | (unsigned)j;
                count++;
            }
        }
    }
    return count;
}

int ms_host_fold_hits(const float* s, const int* line, int n_hits, const int* order, float* nearest_s, int* nearest_line) {
    // one ray's hits through the three slots the way a wave plays them: windows of 64 in the given order, and within a
    // window in lockstep - every hit's first merge, then the second merges of those that go on, then the third
    unsigned long long best = ~0ull, second = ~0ull, third = ~0ull;
    for (int w0 = 0; w0 < n_hits; w0 += WAVE) {
        const int nw = (n_hits - w0 < WAVE) ? n_hits - w0 : WAVE;
        unsigned long long lose1[WAVE], lose2[WAVE];
        bool on[WAVE];
        for (int k = 0; k < nw; k++) {
            const int h = order[w0 + k];
            const unsigned long long key = hit_key(s[h], line[h]);
            on[k] = hit_loser_matters(key, s[h], slot_min(&best, key), lose1[k]);
        }
        for (int k = 0; k < nw; k++) if (on[k]) {
            const unsigned long long old2 = slot_min(&second, lose1[k]);
            lose2[k] = old2 > lose1[k] ? old2 : lose1[k];
        }
        for (int k = 0; k < nw; k++) if (on[k] && lose2[k] != ~0ull) slot_min(&third, lose2[k]);
    }
    *nearest_s = INFINITY; *nearest_line = -1;
    return hit_resolve(best, second, third, *nearest_s, *nearest_line) ? 1 : 0;
}

float ms_host_wall_reach(const float* agent, float agent_radius) { return wall_reach(p2(agent[0], agent[1]), p2(agent[2], agent[3]), agent_radius); }

int ms_host_wall_beyond_reach(const float* agent, const float* wall, float agent_radius) {
    const float reach = wall_reach(p2(agent[0], agent[1]), p2(agent[2], agent[3]), agent_radius);
    return wall_beyond(make_float4(agent[0], agent[1], agent[2], agent[3]), make_float4(wall[0], wall[1], wall[2], wall[3]),
                       reach_squared(reach)) ? 1 : 0;
}

int ms_host_agents_apart(const float* me, const float* other, float agent_radius) {
    return agents_apart(make_float4(me[0], me[1], me[2], me[3]), make_float4(other[0], other[1], other[2], other[3]), agent_radius) ? 1 : 0;
}

int ms_host_wall_hidden(float x0, float y0, float x1, float y1, const float* o, const float* w, float near_plane) {
    const WgCell k{x0, y0, x1, y1};
    const WgTarget t = wg_target(k, make_float4(w[0], w[1], w[2], w[3]));
    return wg_hides(k, t, make_float4(o[0], o[1], o[2], o[3]), near_plane) ? 1 : 0;
}

void ms_host_wallgrid_cell(const float* walls, int n_walls, float ox, float oy, int nx, int ny, float cell, int c,
                           float near_plane, float reach_lo, float reach, unsigned char* vis, unsigned char* close) {
    const float4* ln = reinterpret_cast<const float4*>(walls);
    const WgCell k = wg_cell_of(make_float4(ox, oy, (float)nx, (float)ny), cell, c);
    for (int t = 0; t < n_walls; t++) {
        const WgTarget tg = wg_target(k, ln[t]);
        bool hidden = false;
        for (int o = 0; o < n_walls && !hidden; o++) hidden = (o != t) && wg_hides(k, tg, ln[o], near_plane);
        vis[t] = hidden ? 0 : 1;
        close[t] = wg_close(k, ln[t], reach) ? (wg_close(k, ln[t], reach_lo) ? 2 : 1) : 0;
    }
}

void ms_host_wall_arc(float x0, float y0, float x1, float y1, const float* w, int* lo8, int* hi8) {
    wg_arc(WgCell{x0, y0, x1, y1}, make_float4(w[0], w[1], w[2], w[3]), *lo8, *hi8);
}
int ms_host_wedge_meets(float right_x, float right_y, float left_x, float left_y, int lo8, int hi8) {
    int wa8, wb8;
    wg_wedge(pseudo_angle(right_x, right_y), pseudo_angle(left_x, left_y), wa8, wb8);
    return wg_arcs_meet(lo8, hi8, wa8, wb8) ? 1 : 0;
}

static bool wallgrid_ok(const MsScenery* sc) {
    return sc->wg_starts && sc->wg_geom && sc->wg_cell > 0.f && sc->wg_reach_lo >= 0.f && sc->wg_reach >= sc->wg_reach_lo &&
           sc->wg_near > 0.f && ((uintptr_t)sc->wg_geom % 16 == 0);
}

int ms_wallgrid_scan(const MsScenery* sc, const MsWallGridParent* parent, const int* reps, int n_reps, int max_groups,
                     const long long* bits_starts, unsigned* bits, unsigned* counts, void* stream) {
    if (!scenery_ok(sc) || !wallgrid_ok(sc) || !reps || n_reps < 0 || max_groups < 0 || !bits_starts || !bits || !counts) return MS_EINVAL;
    WgParent par{nullptr, nullptr, nullptr, 0.f, nullptr};
    if (parent) {
        if (!parent->cells || !parent->starts || !parent->geom || !parent->pool || !(parent->cell >= sc->wg_cell) ||
            ((uintptr_t)parent->cells % 16) || ((uintptr_t)parent->geom % 16)) return MS_EINVAL;
        par = WgParent{parent->cells, parent->starts, parent->geom, parent->cell, parent->pool};
    }
    if (n_reps == 0 || max_groups == 0) return MS_OK;
    if (n_reps > 65535) return MS_EUNSUPPORTED;
    hipLaunchKernelGGL(wallgrid_scan_kernel, dim3((unsigned)max_groups, (unsigned)n_reps), dim3(WG), 0, (hipStream_t)stream,
                       *sc, par, reps, bits_starts, bits, counts);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MS_OK : hip_fail(e);
}

int ms_wallgrid_fill(const MsScenery* sc, const int* reps, int n_reps, int max_cells,
                     const long long* bits_starts, const unsigned* bits, unsigned short* pool, unsigned* vis_entries, float* near_rows,
                     void* stream) {
    if (!scenery_ok(sc) || !wallgrid_ok(sc) || !sc->wg_cells || ((uintptr_t)sc->wg_cells % 16) || !reps || n_reps < 0 || max_cells < 0 ||
        !bits_starts || !bits || ((vis_entries != nullptr) != (near_rows != nullptr)) || (!pool && !vis_entries) ||
        ((uintptr_t)near_rows % 16) || ((uintptr_t)vis_entries % 4)) return MS_EINVAL;
    if (n_reps == 0 || max_cells == 0) return MS_OK;
    const long long blocks = (2LL*max_cells + WAVES - 1)/WAVES;
    if (blocks > 0x7fffffffLL || n_reps > 65535) return MS_EUNSUPPORTED;
    hipLaunchKernelGGL(wallgrid_fill_kernel, dim3((unsigned)blocks, (unsigned)n_reps), dim3(WG), 0, (hipStream_t)stream,
                       *sc, reps, bits_starts, bits, pool, reinterpret_cast<float4*>(near_rows), vis_entries);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MS_OK : hip_fail(e);
}

int ms_step_physics(const MsScenery* sc, const MsAgents* ag, const MsMovement* mv, const MsStepExtras* ex, float* progress,
                    const MsConfig* cfg, void* stream) {
    if (!scenery_ok(sc) || !agents_ok(ag) || !progress || !config_ok(cfg)) return MS_EINVAL;
    if (mv && (!mv->actions || !mv->table || mv->n_actions < 1 || !(mv->keep == mv->keep))) return MS_EINVAL;
    if (sc->wg_cells && (!sc->wg_starts || !sc->wg_geom || !sc->wg_near_rows || !(sc->wg_cell > 0.f) || ((uintptr_t)sc->wg_cells % 16) ||
                         ((uintptr_t)sc->wg_geom % 16) || ((uintptr_t)sc->wg_near_rows % 16))) return MS_EINVAL;
    if (ex) {
        if (ex->spawn_positions && (!ex->spawn_angles || !ex->respawn_mask || !ex->respawn_choice || ex->n_spawns < 1 ||
                                    ((uintptr_t)ex->spawn_positions % 8))) return MS_EINVAL;
        if (ex->lifespans && (!ex->max_lifespans || !ex->fresh_max)) return MS_EINVAL;
        if (ex->imu && !(ex->imu_ang_scale == ex->imu_ang_scale && ex->imu_speed_scale == ex->imu_speed_scale)) return MS_EINVAL;
    }
    // per env: 2 float4 + a float + an unsigned per agent, rounded up to whole float4s
    const size_t slice = ((sizeof(float)*8 + sizeof(float) + sizeof(unsigned))*(size_t)sc->n_agents + 15)/16;
    if (slice*16 > 56*1024) return MS_EUNSUPPORTED;
    const MsMovement no_move{nullptr, nullptr, 0, 0.f};
    const MsStepExtras no_extras{nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, 1.f, 1.f};
    const MsMovement mvv = mv ? *mv : no_move;
    const MsStepExtras exv = ex ? *ex : no_extras;
    MsScenery scn = *sc;
    if (!sc->wg_cells) { scn.wg_geom = sc->lines_vals; scn.wg_starts = sc->lines_starts; }   // (rows the kernel may read: see there)
    const hipStream_t hs = (hipStream_t)stream;
    // one wavefront per env (several envs per wave, one after the other: 2 -> +25 %, 4 -> +85 % at 4096 envs)
#define MS_LAUNCH_PHYSICS(M, E) \
    hipLaunchKernelGGL((physics_kernel<M, E>), dim3(sc->n_envs), dim3(WAVE), slice*16, hs, scn, *ag, progress, cfg->agent_radius, cfg->fps, mvv, exv)
    if (mv && ex) MS_LAUNCH_PHYSICS(1, 1);
    else if (ex) MS_LAUNCH_PHYSICS(0, 1);
    else if (mv) MS_LAUNCH_PHYSICS(1, 0);
    else MS_LAUNCH_PHYSICS(0, 0);
#undef MS_LAUNCH_PHYSICS
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MS_OK : hip_fail(e);
}

int ms_move_physics(const MsScenery* sc, const MsAgents* ag, const MsMovement* mv, float* progress, const MsConfig* cfg,
                    void* stream) {
    return ms_step_physics(sc, ag, mv, nullptr, progress, cfg, stream);
}

int ms_physics(const MsScenery* sc, const MsAgents* ag, float* progress, const MsConfig* cfg, void* stream) {
    return ms_step_physics(sc, ag, nullptr, nullptr, progress, cfg, stream);
}

int ms_render(const MsScenery* sc, const MsAgents* ag, const MsRender* out, const MsConfig* cfg, void* stream) {
    if (!scenery_ok(sc) || !agents_ok(ag) || !config_ok(cfg) || !out || !sc->textures_vals || !sc->textures_widths ||
        !sc->textures_starts || !sc->baked_vals || !sc->lights_widths || !sc->lights_starts) return MS_EINVAL;
    if ((out->seen_stamp != nullptr) != (out->seen_epoch != nullptr) || (out->seen_stamp != nullptr) != (out->seen_count != nullptr)) return MS_EINVAL;
    if (out->obs_rgb || out->obs_depth || out->obs_centre) {
        const int sub = out->obs_subsample;
        if (sub < 1 || (sub & (sub - 1)) || sub > WAVE || cfg->res % sub) return MS_EINVAL;
        if (out->obs_depth && !(out->obs_max_depth > 0.f)) return MS_EINVAL;
        if (out->obs_centre && cfg->res/sub < 2) return MS_EINVAL;
    }
    if (sc->n_lights_total > 0 && !sc->lights_vals) return MS_EINVAL;
    const int R = cfg->res;
    // ray groups per wave (render_kernel's NG): an agent's groups of 64 rays share the wave's list of walls instead of each
    // wave building its own.  (ms_debug_ray_groups: A/B runs and tests pin it.)
    // Measured: four groups pay from 256 rays up - a quarter of the vector instructions saved without colour, a
    // sixth with - IF the launch has two and a half rounds of such waves to fill the machine with and its last envs are left to
    // waves of one group (below): 4096 x 4 x 512 rays 157.5 -> 146.7 us (colourless 127.4 -> 110.4), C5's share
    // 216.7 -> 207.9 (152.8 -> 132.2); with a round or less of them - 4096 x 1 x 256 rays - a quarter is LOST.  Two groups
    // never pay.
    static int slots = 0;                                                // the machine's wave slots for this kernel
    if (!slots) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        slots = cus*4*MS_WAVES;
    }
    int ng = 1;
    if (R >= 4*WAVE && 2LL*sc->n_envs*sc->n_agents*((R + 4*WAVE - 1)/(4*WAVE)) >= 5LL*slots) ng = 4;   // (2.5 rounds, the one-group waves' half included)
    if (g_ray_groups == 1 || g_ray_groups == 2 || g_ray_groups == 4) ng = g_ray_groups;
#if MS_AB_IMPLS
    if (getenv("MEGASTEP_RENDER_IMPL")) ng = 1;
#endif
    // (without a light grid the rays that land on an agent are lit by dynlight_kernel, which takes them by groups of 64)
    if (!(sc->lg_vals && sc->lg_starts && sc->lg_geom && sc->lg_cell > 0.f) && sc->n_agents > 1 && (out->screen || out->obs_rgb)) ng = 1;
    // (NG > 1: waves of ng groups for every XCD's envs but its last `tail`, waves of one group for those - render_kernel has
    // why.  Their share: g_tail_rounds rounds of the machine's wave slots' worth of the wide waves' work, half a round unless
    // ms_debug_ray_group_tail says otherwise.)
    int tail = 0;
    const int G1 = (R + WAVE - 1)/WAVE;
    const int envs_lo = sc->n_envs/8, envs_rem = sc->n_envs % 8, envs_hi = envs_lo + (envs_rem ? 1 : 0);
    if (ng > 1) {
        const double rounds = g_tail_rounds >= 0.f ? (double)g_tail_rounds : 0.5;
        const long long tail_envs = g_tail_envs >= 0 ? g_tail_envs : (long long)ceil(rounds*slots*ng/((double)sc->n_agents*G1));
        tail = (int)std::min<long long>((tail_envs + 7)/8, envs_hi);                        // per XCD
        if (tail >= envs_hi && !(g_ray_groups > 1)) ng = 1;           // nothing left for the wide waves: the plain kernel
    }
    const int G = (R + ng*WAVE - 1)/(ng*WAVE);
    // (NG > 1: every XCD as many blocks as the one with the most envs needs)
    const long long n_fans = ng > 1 ? 8LL*((long long)(envs_hi - std::min(tail, envs_hi))*sc->n_agents*G + (long long)std::min(tail, envs_hi)*sc->n_agents*G1)
                                    : (long long)sc->n_envs*sc->n_agents*G;
    if (n_fans > 0x7fffffffLL) return MS_EUNSUPPORTED;
    // kernels.cu:22
    const float half_screen = tanf(3.14159265358979323846f/180.f*cfg->fov/2.);
#if MS_AB_IMPLS
    // MEGASTEP_RENDER_IMPL: "seq" (literal order, slowest), "pairs" (round 1's pair raycast), anything else the product's
    // kernel; all three produce the same bits.  Read per call: tests switch it.
    const char* impl_env = getenv("MEGASTEP_RENDER_IMPL");
    const bool seq = impl_env && impl_env[0] == 's';
    const bool pairs1 = impl_env && impl_env[0] == 'p';
#endif
    // the light grid is all or nothing: render_kernel lights agent-hit rays itself when it is there
    MsScenery scn = *sc;
    const bool grid = sc->lg_vals && sc->lg_starts && sc->lg_geom && sc->lg_cell > 0.f;
    if (!grid) scn.lg_vals = nullptr;
    // ... and so is the wall grid: its vis lists were built for near planes below wg_near and ray direction vectors no
    // longer than sqrt(WG_MAX_RU2) (wallgrid_scan_kernel); a call outside that meets every wall instead
    const bool walls_listed = sc->wg_cells && sc->wg_starts && sc->wg_geom && sc->wg_pool && sc->wg_cell > 0.f &&
                              cfg->agent_radius*1.001f < sc->wg_near && 1.f + half_screen*half_screen <= WG_MAX_RU2;
    if (walls_listed && (((uintptr_t)sc->wg_cells % 16) || ((uintptr_t)sc->wg_geom % 16))) return MS_EINVAL;
    if (!walls_listed) {                                                 // (the kernel reads a row of each whatever happens: see there)
        scn.wg_cells = nullptr;
        scn.wg_geom = sc->lines_vals;                                    // at least 16 bytes per env: every env has its agents' lines
        scn.wg_starts = sc->lines_starts;
    }
    // Headings: from ms_physics' cache when the agents carry one and a single kernel does the whole job (then the
    // workspace is not needed at all); otherwise from render_prep_kernel, which also resets the workspace's counters.
    MsAgents agn = *ag;
    MsRender outn = *out;
    const bool colour = out->screen || out->obs_rgb;                      // else: render_kernel<.,.,1,0>, which has no pass 3
    const bool one_kernel = grid || sc->n_agents == 1 || !colour;        // (nothing to light without colour)
    if (ag->headings && one_kernel) {
        if ((uintptr_t)ag->headings % 16) return MS_EINVAL;
        outn.workspace = nullptr;
    } else {
        agn.headings = nullptr;
        if (out->workspace) {
            if ((uintptr_t)out->workspace % 8) return MS_EINVAL;
            const int na = sc->n_envs*sc->n_agents;
            hipLaunchKernelGGL(render_prep_kernel, dim3((na + WG - 1)/WG), dim3(WG), 0, (hipStream_t)stream,
                               *ag, out->workspace, na, (int)n_fans);
        }
    }
    // dynlight_kernel reads the per-ray outputs back: only the one-kernel path can do without some of them
    const bool all_planes = out->indices && out->locations && out->dots && out->distances && out->screen;
    const bool pooled = out->obs_rgb || out->obs_depth || out->obs_centre || out->seen_stamp;
    if (colour && (!all_planes || pooled) && !(grid || sc->n_agents == 1)) return MS_EUNSUPPORTED;   // dynlight_kernel patches `screen` afterwards
    if (!all_planes && !pooled && !out->indices && !out->locations && !out->dots && !out->distances && !out->screen) return MS_EINVAL;
    const bool obs = pooled || !all_planes;
    RenderConsts rc;
    rc.x_clip = 0.5f*cfg->agent_radius/sqrtf(1.f + half_screen*half_screen);
    rc.c_b = 0.5f*(float)R/half_screen;
    rc.by_f = divisor_of((unsigned)(sc->n_agents*G));
    rc.by_g = divisor_of((unsigned)G);
    rc.by_m = divisor_of((unsigned)sc->n_model);
    rc.by_f1 = divisor_of((unsigned)(sc->n_agents*G1));
    rc.by_g1 = divisor_of((unsigned)G1);
    rc.envs_lo = envs_lo; rc.envs_rem = envs_rem; rc.tail = tail;
    rc.skip_own = (sc->model_radius > 0.f && sc->model_radius*1.01f < cfg->agent_radius) ? 1 : 0;
    rc.inv_res = ((R & (R - 1)) == 0 && half_screen > 1e-3f) ? 1.f/(float)R : 0.f;
    rc.telemetry = g_pair_telemetry;
    constexpr int RW = 1;
    const int rblocks = (int)((n_fans + RW - 1)/RW);
    const dim3 rgrid(rblocks), rblock(RW*WAVE);
    const hipStream_t hs = (hipStream_t)stream;
#define MS_LAUNCH_RENDER(I, O) \
    hipLaunchKernelGGL((render_kernel<I, RW, O, 1>), rgrid, rblock, 0, hs, scn, agn, outn, cfg->agent_radius, half_screen, R, (int)n_fans, rc)
#if MS_AB_IMPLS
    if (seq) { if (obs) MS_LAUNCH_RENDER(0, 1); else MS_LAUNCH_RENDER(0, 0); }
    else if (pairs1) { if (obs) MS_LAUNCH_RENDER(1, 1); else MS_LAUNCH_RENDER(1, 0); }
    else
#endif
#define MS_LAUNCH_RENDER_NG(O, S, NG_) \
    hipLaunchKernelGGL((render_kernel<2, RW, O, S, NG_>), rgrid, rblock, 0, hs, scn, agn, outn, cfg->agent_radius, half_screen, R, (int)n_fans, rc)
#define MS_LAUNCH_RENDER_OS(NG_) \
    { if (!colour) MS_LAUNCH_RENDER_NG(1, 0, NG_); else if (obs) MS_LAUNCH_RENDER_NG(1, 1, NG_); else MS_LAUNCH_RENDER_NG(0, 1, NG_); }
    if (ng == 4) MS_LAUNCH_RENDER_OS(4)
    else if (ng == 2) MS_LAUNCH_RENDER_OS(2)
    else MS_LAUNCH_RENDER_OS(1)
#undef MS_LAUNCH_RENDER_OS
#undef MS_LAUNCH_RENDER_NG
#undef MS_LAUNCH_RENDER
    // without a grid: second launch.  With one agent per env no ray can land on an agent line (own lines sit
    // inside the near plane), so there is nothing to light.
    if (!grid && sc->n_agents > 1 && colour)
        hipLaunchKernelGGL(dynlight_kernel, dim3((int)n_fans), dim3(WG), 0, (hipStream_t)stream, scn, *ag, *out, R);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MS_OK : hip_fail(e);
}

int ms_bake(const MsScenery* sc, const MsConfig* cfg, void* stream) {
    (void)cfg;
    if (!scenery_ok(sc) || !sc->textures_widths || !sc->textures_starts || !sc->textures_inverse ||
        !sc->baked_vals || !sc->lights_widths || !sc->lights_starts) return MS_EINVAL;
    if (sc->n_lights_total > 0 && !sc->lights_vals) return MS_EINVAL;
    if (sc->n_texels_total > 0 && sc->bake_vis) {
        // two phases: visibility once per representative env and light, then the per-env sums
        if (!sc->bake_vis_starts || sc->bake_vis_words < 0 || !sc->lines_inverse || ((uintptr_t)sc->bake_vis % 8)) return MS_EINVAL;
        const char* be = getenv("MEGASTEP_BAKE_BINS");                 // =0: every texel meets every wall (A/B runs)
        const bool bins = !(be && be[0] == '0');
        if (sc->n_lights_total > 0)
            hipLaunchKernelGGL(visibility_kernel, dim3(sc->n_lights_total), dim3(WG), 0, (hipStream_t)stream, *sc, bins ? 1 : 0);
        const long long blocks = ((long long)sc->n_texels_total + WG - 1)/WG;
        hipLaunchKernelGGL(bake_sum_kernel, dim3((unsigned)blocks), dim3(WG), 0, (hipStream_t)stream, *sc);
    } else if (sc->n_texels_total > 0) {
        hipLaunchKernelGGL(bake_kernel, dim3(sc->n_envs), dim3(WG), 0, (hipStream_t)stream, *sc);
    }
    if (sc->lg_vals) {
        if (!sc->lg_starts || !sc->lg_geom || !(sc->lg_cell > 0.f) || sc->lg_max_cells <= 0 || ((uintptr_t)sc->lg_vals % 16) ||
            ((uintptr_t)sc->lg_geom % 16)) return MS_EINVAL;
        if ((sc->lg_list != nullptr) != (sc->lg_pool != nullptr) || (sc->lg_pool && sc->lg_pool_size < 1) ||
            (sc->lg_pool_rows && (!sc->lg_pool || ((uintptr_t)sc->lg_pool_rows % 16))) ||
            ((uintptr_t)sc->lg_list % 8)) return MS_EINVAL;
        const dim3 cells((sc->lg_max_cells + WG - 1)/WG, sc->n_envs);
        hipLaunchKernelGGL(lightgrid_kernel, cells, dim3(WG), 0, (hipStream_t)stream, *sc);
        if (sc->lg_list) {
            if (hipMemsetAsync(sc->lg_pool, 0, sizeof(unsigned), (hipStream_t)stream) != hipSuccess) return hip_fail(hipGetLastError());
            hipLaunchKernelGGL(lightlist_kernel, cells, dim3(WG), 0, (hipStream_t)stream, *sc);
        }
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MS_OK : hip_fail(e);
}

}  // extern "C"
