// kernels/wallgrid.h -- wallgrid_scan_kernel, wallgrid_fill_kernel.
// Part of megastep_hip.hip's one translation unit (included there, inside its anonymous namespace, in this order: math,
// physics, lighting, render, bake, wallgrid); not a header to compile on its own.
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
// Round 6: which occluders a target wall is tried against.  Until then a wall that stays visible met every staged occluder - a
// hundred of them on a large plan, for two hundred candidates, for each of 51 million cells: 16 of the 17 seconds C5's share takes
// to build (profiles/r05_c5_kernel_stats.csv).  But O can only hide W from the cell if every segment from the cell to W crosses it
// (above) - the one from the cell's centre to W's middle among them: O must SPAN THE DIRECTION in which W's middle lies from the
// cell's centre.  So per cell the occluders are sorted into WG_SECTORS sectors of directions around its centre - each into every
// sector its own span touches, widened by WG_SECTOR_MARGIN (in pseudo-angle units: a full turn is 4; the margin is 10^4 roundings
// wide) - and a target meets the occluders of the ONE sector its middle lies in: half a dozen.  An exact cull of the loop, not of
// the lists: the bits come out as they did (tests/test_gpu_wallgrid.py holds them against the host's unsorted scan).  Lists that
// overflow WG_SECTOR_CAP (a cell hemmed in by hundreds of long walls) fall back to meeting every occluder.
constexpr int WG_SECTORS = 64, WG_SECTOR_CAP = 4096;
constexpr float WG_SECTOR_MARGIN = 0.02f;
// the run of sectors [first, first + count) (modulo WG_SECTORS) that wall o's directions from (cx, cy) touch; all of them where the
// span is half a turn or more, or undefined (a wall through the centre, NaNs)
__host__ __device__ inline void wg_sectors_of(const float cx, const float cy, const float4 o, int& first, int& count) {
    const float pa = pseudo_angle(o.x - cx, o.y - cy), pb = pseudo_angle(o.z - cx, o.w - cy);
    float d = pb - pa;                                                   // from a to b, the short way round
    d = d > 2.f ? d - 4.f : (d <= -2.f ? d + 4.f : d);
    const float lo = (d >= 0.f ? pa : pb) - WG_SECTOR_MARGIN, span = fabsf(d) + 2.f*WG_SECTOR_MARGIN;
    first = 0; count = WG_SECTORS;
    if (!(span < 1.9f) || !(lo == lo)) return;
    const float scale = WG_SECTORS/4.f;
    const int s0 = (int)floorf(lo*scale), s1 = (int)floorf((lo + span)*scale);
    first = s0 & (WG_SECTORS - 1);
    count = min(s1 - s0 + 1, WG_SECTORS);
}
__host__ __device__ inline int wg_sector_of(const float cx, const float cy, const float4 w) {
    const float p = pseudo_angle(.5f*(w.x + w.z) - cx, .5f*(w.y + w.w) - cy);
    return (p == p) ? ((int)floorf(p*(WG_SECTORS/4.f)) & (WG_SECTORS - 1)) : 0;   // (a NaN wall is hidden by nothing: any sector will do)
}

struct WgParent { const unsigned* cells; const int* starts; const float* geom; float cell; const unsigned short* pool; };

__global__ __launch_bounds__(WG) void wallgrid_scan_kernel(const MsScenery sc, const WgParent parent, const int* __restrict__ reps,
                                                          const long long* __restrict__ bits_starts, unsigned* __restrict__ bits,
                                                          unsigned* __restrict__ counts) {
    __shared__ float4 s_occ[WG_STAGE];
    __shared__ unsigned short s_occ_id[WG_STAGE];
    __shared__ int s_n_occ;
    __shared__ int s_sect_at[WG_SECTORS + 1], s_sect_fill[WG_SECTORS];   // where each sector's list starts (an exclusive scan: [s] .. [s + 1]), how far it is filled
    __shared__ unsigned short s_sect[WG_SECTOR_CAP];
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
    // vis: cell after cell of the group - its occluders sorted into sectors of directions (see WG_SECTORS), then lane = candidate,
    // 64 to a wave, each against the occluders of the sector its middle lies in
    const int vis_chunks = (n_vis + WAVE - 1)/WAVE;
#ifndef MS_WG_SECTORS
#define MS_WG_SECTORS 1                                                  // (0: every target meets every occluder, as until round 6 - the A/B)
#endif
    for (int j = 0; j < n_group; j++) {
        const int c = cell_of(j);
        const WgCell k = wg_cell_of(geom, sc.wg_cell, c);
        const float cx = .5f*(k.x0 + k.x1), cy = .5f*(k.y0 + k.y1);
        bool sorted = false;
        if (MS_WG_SECTORS && n_occ > 16) {
            if (tid <= WG_SECTORS) s_sect_at[tid] = 0;
            if (tid < WG_SECTORS) s_sect_fill[tid] = 0;
            __syncthreads();
            for (int o = tid; o < n_occ; o += WG) {
                int first, count;
                wg_sectors_of(cx, cy, s_occ[o], first, count);
                for (int t = 0; t < count; t++) atomicAdd(&s_sect_at[1 + ((first + t) & (WG_SECTORS - 1))], 1);
            }
            __syncthreads();
            if (wave == 0) {                                             // counts -> where each sector's list starts
                const int n_ = s_sect_at[1 + lane];
                const int incl = wave_scan_add(n_);
                __builtin_amdgcn_wave_barrier();
                s_sect_at[1 + lane] = incl;
            }
            __syncthreads();
            sorted = s_sect_at[WG_SECTORS] <= WG_SECTOR_CAP;              // (uniform: every thread reads the same word)
            if (sorted) {
                for (int o = tid; o < n_occ; o += WG) {
                    int first, count;
                    wg_sectors_of(cx, cy, s_occ[o], first, count);
                    for (int t = 0; t < count; t++) {
                        const int sct = (first + t) & (WG_SECTORS - 1);
                        s_sect[s_sect_at[sct] + atomicAdd(&s_sect_fill[sct], 1)] = (unsigned short)o;
                    }
                }
            }
            __syncthreads();
        }
        for (int chunk = wave; chunk < vis_chunks; chunk += WAVES) {
            const int i = chunk*WAVE + lane;
            const bool live = i < n_vis;
            const int id = cand_vis ? (int)cand_vis[min(i, n_vis - 1)] : min(i, n_vis - 1);
            const float4 w = ln[id];
            const WgTarget tg = wg_target(k, w);
            bool hidden = !live;
            if (sorted) {
                const int sct = wg_sector_of(cx, cy, w);
                int q = s_sect_at[sct];
                const int q_end = s_sect_at[sct + 1];
                while (__ballot(!hidden & (q < q_end))) {
                    if (!hidden & (q < q_end)) {
                        const int o = (int)s_sect[q];
                        if (((int)s_occ_id[o] != id) && wg_hides(k, tg, s_occ[o], sc.wg_near)) hidden = true;
                    }
                    q++;
                }
            } else {
                for (int o = 0; o < n_occ; o++) {
                    if (__all(hidden)) break;
                    if (((int)s_occ_id[o] != id) && wg_hides(k, tg, s_occ[o], sc.wg_near)) hidden = true;
                }
            }
            if (!hidden) atomicOr(&rows[(long long)(WG_ROWS*c)*W32 + (id >> 5)], 1u << (id & 31));
        }
        if (MS_WG_SECTORS && n_occ > 16) __syncthreads();                 // (the next cell sorts into the same lists)
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
    // (the vis lists of the final level count from the floorplan's own place in the pool: MsScenery.wg_pool_base)
    if (vis_entries) vis_entries += sc.wg_pool_base[n];
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
