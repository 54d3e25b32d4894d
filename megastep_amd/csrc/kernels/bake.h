// kernels/bake.h -- bake_kernel, visibility_kernel, bake_sum_kernel, lightgrid_kernel, lightlist_kernel.
// Part of megastep_hip.hip's one translation unit (included there, inside its anonymous namespace, in this order: math,
// physics, lighting, render, bake, wallgrid); not a header to compile on its own.
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
