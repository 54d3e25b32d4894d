// kernels/math.h -- scalar math and the small helpers the kernels share (and ms_host_* instantiate on the host).
// Part of megastep_hip.hip's one translation unit (included there, inside its anonymous namespace, in this order: math,
// physics, lighting, render, bake, wallgrid); not a header to compile on its own.
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

// n / d where the quotient needs none of the range handling the compiler's expansion of a correctly rounded division carries:
// that expansion is v_div_scale x 2 (which move operands near the ends of the exponent range towards the middle: a denormal
// or > 2^126 divisor, a numerator below 2^-104, exponents more than 96 apart), v_rcp_f32, the Newton-Raphson steps below, and
// v_div_fmas / v_div_fixup (which undo the scaling and patch zeros, infinities and NaNs in).  For operands in range the scaling
// is the identity and the fix-up a move: the SAME steps on the SAME values, hence the same bits - the correctly rounded
// quotient (Markstein's sequence) - for 8 instructions instead of 11, at ten divisions a wave (6 % of a render wave's vector
// instructions with the shared reciprocal of tex_filter and the square root below).  In range by construction at every site
// it is used: divisors are |cross(ray, wall)| >= 1e-3 of a hit, a ray's length in [1, 12], texel weights' sums >= 2e-3, a
// wall's length + 1e-6; numerators of hits that can be taken are >= 1e-5.  Where the operands leave the range - a numerator that
// is zero or below 2^-104, coordinates beyond 10^15 - the quotient is a float output within its 1e-5 tolerance (zero, or a few
// ulps of something tiny) or the s of a hit inside the near plane that the comparison throws away either way; never a hit index.
#ifndef MS_DIV_INRANGE
#define MS_DIV_INRANGE 1               // (0: plain `/` everywhere - the A/B, and the same bits)
#endif
__device__ inline float rcp_refined(const float d) {                   // 1/d to within an ulp: the reciprocal the steps below share
    const float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.f), r, r);
}
__device__ inline float div_by_refined(const float n, const float d, const float r) {   // n / d given r = rcp_refined(d)
    float q = n*r;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
    return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
}
__device__ inline float div_inrange(const float n, const float d) {
#if MS_DIV_INRANGE
    return div_by_refined(n, d, rcp_refined(d));
#else
    return n/d;
#endif
}

// Exact unsigned division by a divisor the host knows before the launch (agents per env, ray groups per agent, lines per agent):
// multiply-high and two shifts (Granlund & Montgomery) - a run-time integer division costs a wave thirty vector instructions.
struct Divisor { unsigned mul, sh1, sh2; };
__host__ inline Divisor divisor_of(unsigned d) {           // d >= 1
    unsigned s = 0;
    while ((1ull << s) < d) s++;
    const unsigned long long m = ((1ull << 32)*((1ull << s) - d))/d + 1ull;
    return Divisor{(unsigned)m, s < 1u ? s : 1u, s > 1u ? s - 1u : 0u};
}
__host__ __device__ inline int div_by(int n, const Divisor d) {     // n >= 0
    const unsigned t = (unsigned)(((unsigned long long)d.mul*(unsigned)n) >> 32);     // (the high word: one v_mul_hi_u32 / s_mul_hi_u32)
    return (int)((t + (((unsigned)n - t) >> d.sh1)) >> d.sh2);
}

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
