// kernels/physics.h -- physics_kernel<MOVE, EXTRA>.
// Part of megastep_hip.hip's one translation unit (included there, inside its anonymous namespace, in this order: math,
// physics, lighting, render, bake, wallgrid); not a header to compile on its own.
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
// PACK = 1: a wave takes `pack_envs` consecutive envs side by side (ms_step_physics: worlds of several rounds of waves with a
//   wall grid and few agents per env - 32768 envs of one agent are 5.3 rounds of waves with a lane or two at work; eight envs
//   to a wave they are two thirds of one).  The (N, A) arrays are row-major, so the wave's agents are consecutive rows of
//   every one of them: lane = agent as before, `A` the wave's agents (pack_envs x n_agents <= 64), and only what is per env -
//   who can run into whom, whose cell lists, which env falls back to meeting all its walls - looks at lane / n_agents.
template <int MOVE, int EXTRA, int PACK = 0>
__global__ __launch_bounds__(WAVE) void physics_kernel(
        const MsScenery sc, const MsAgents ag, float* __restrict__ progress,
        const float agent_radius, const float fps, const MsMovement mv, const MsStepExtras ex, const int pack_envs, const Divisor by_a) {
    PROBE_INIT
    extern __shared__ float4 s_dyn[];            // per agent: (p, v/fps) | reach box | reach^2 | progress bits
    __shared__ float4 s_wall[PHYS_PAIRS];        // walls near ...
    __shared__ int s_tag[PHYS_PAIRS];            // ... this agent
    const int A1 = sc.n_agents, AF = sc.n_agents*sc.n_model;            // agents per env
    const int lane = threadIdx.x;
    const int n = PACK ? blockIdx.x*pack_envs : blockIdx.x;              // the host launches one wave per env (PACK: per pack_envs envs; n: the first)
    const int E = PACK ? min(pack_envs, sc.n_envs - n) : 1;
    const int A = PACK ? A1*E : A1;                                      // the wave's agents: rows nA .. nA + A - 1 of every (N, A) array
    const int nA = n*A1;
    // (by_a: division by the agents per env, worked out by the host - `lane / A1` sat in front of the wave's first loads)
    [[maybe_unused]] const int lane_env = PACK ? min(div_by(lane, by_a), E - 1) : 0;
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
    const float4 wg_geom_n = reinterpret_cast<const float4*>(sc.wg_geom)[n + lane_env];   // (PACK: the lane's env's, not the wave's)
    const int wg_start_n = sc.wg_starts[n + lane_env];
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
    if (lane < A) { my_p = pos2[nA + lane]; my_v = vel2[nA + lane]; my_w = ag.angvelocity[nA + lane]; my_ang = ag.angles[nA + lane]; }

    // the spawn pose of agent i, if it is to be respawned (modules.py:321-326)
    auto spawn_pose = [&](const int i, float2& p, float& ang) {
        const long long c = min(max(ex.respawn_choice[i], 0ll), (long long)ex.n_spawns - 1);
        p = reinterpret_cast<const float2*>(ex.spawn_positions)[(size_t)i*ex.n_spawns + c];
        ang = ex.spawn_angles[(size_t)i*ex.n_spawns + c];
    };
    if constexpr (EXTRA == 1) {
        for (int t = lane; t < A; t += WAVE) {
            const int i = nA + t;
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
            const long long act = min(max(mv.actions[nA + min(lane, A - 1)], 0ll), (long long)mv.n_actions - 1);
            float dx, dy, dw;
            if (small_table) { dx = __shfl(tab, 3*(int)act, WAVE); dy = __shfl(tab, 3*(int)act + 1, WAVE); dw = __shfl(tab, 3*(int)act + 2, WAVE); }
            else { dx = mv.table[3*act]; dy = mv.table[3*act + 1]; dw = mv.table[3*act + 2]; }
            if (lane < A) moved(nA + lane, my_ang, my_v, my_w, dx, dy, dw);
        }
        for (int t = lane + WAVE; t < A; t += WAVE) {                   // agents beyond the first 64: through memory
            float2 v = vel2[nA + t];
            float w = ag.angvelocity[nA + t];
            const long long act = min(max(mv.actions[nA + t], 0ll), (long long)mv.n_actions - 1);
            moved(nA + t, ag.angles[nA + t], v, w, mv.table[3*act], mv.table[3*act + 1], mv.table[3*act + 2]);
        }
    }
    float4 my_box = make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);   // (no agent: a box no finite wall touches)
    float my_reach = 0.f;
    for (int t = lane; t < A; t += WAVE) {
        const float2 pp = (t == lane) ? my_p : pos2[nA + t], mm = (t == lane) ? my_v : vel2[nA + t];
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
    for (int i = lane; i < A*A1; i += WAVE) {                            // (t, one of its env's agents)
        const int t = div_by(i, by_a), d1 = (PACK ? div_by(t, by_a)*A1 : 0) + i - t*A1;
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
        const unsigned long long uncovered = __ballot(!ok);
        [[maybe_unused]] const unsigned long long env_lanes = A1 >= WAVE ? ~0ull : (1ull << A1) - 1ull;
        if constexpr (PACK == 1) {
            // (an env with an agent its lists do not cover meets all its walls, further down; the wave's other envs go by theirs)
            if (uncovered & (env_lanes << (lane_env*A1))) count = 0;
        }
        if (PACK == 1 || !uncovered) {
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
            if constexpr (PACK == 1) {
                for (int e = 0; e < E; e++) {
                    if (!((uncovered >> (e*A1)) & env_lanes)) continue;
                    // (rare, and plain: every wall of the env to every agent of it - meet() has the reach cull on the true
                    // distance in front of the exact test, which is all the sweep's boxes stand in for)
                    const int Le = sc.lines_widths[n + e];
                    const LineRows rows_e(reinterpret_cast<const float4*>(sc.lines_vals) + sc.lines_starts[n + e], Le);
                    for (int l0 = AF; l0 < Le; l0 += WAVE) {
                        const float4 u = rows_e.chunk(lane, l0);
                        if (l0 + lane < Le)
                            for (int t = e*A1; t < (e + 1)*A1; t++) meet(u, t);
                    }
                }
            }
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
        const int i = nA + t;
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
                // (times the reciprocals ms_step_physics left in the two fields: the modules divide a tensor by a Python scalar, which
                // ATen evaluates as `a * (1.f/b)` - torch's own bits)
                ex.imu[3*i] = w_*ex.imu_ang_scale;
                ex.imu[3*i + 1] = (c_*v.x + s_*v.y)*ex.imu_speed_scale;
                ex.imu[3*i + 2] = (-s_*v.x + c_*v.y)*ex.imu_speed_scale;
            }
        }
    }
    PROBE_DONE(n)
}
