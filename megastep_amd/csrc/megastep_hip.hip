// megastep_hip.hip -- gfx950 (MI355X / CDNA4) simulation core behind include/megastep_hip.h.
//
// One translation unit: this file holds the switches, the probe macros and the host side of the C-ABI (argument checks,
// launches); the kernels are in kernels/*.h, included below inside the anonymous namespace - math.h (scalar helpers shared
// with the ms_host_* test hooks), physics.h, lighting.h, render.h, bake.h, wallgrid.h.
//
// Thirteen kernels, all written wave64-first (DESIGN.md section 3 has the full story of each):
//
//   physics_kernel<MOVE, EXTRA, PACK>   one wavefront per env (PACK = 1: per few consecutive envs, side by side - large
//                   worlds of few agents per env): lane = agent for the state, the reach and the agent-agent
//                   tests; the walls from the near lists of the agents' cells of the wall grid (rows of the dozen walls
//                   within reach, dealt to the lanes one (wall, agent) pair each) - or, where no list applies, a sweep
//                   over all the env's walls (buffer loads in flight, lane = wall, reach boxes in scalar registers,
//                   compacted pairs); the exact collision test, atomicMin fold, integration epilogue; leaves each
//                   agent's sin/cos for the renderer.  MOVE = 1 runs the movement modules' velocity update first,
//                   EXTRA = 1 the envs' respawn / lifespan / IMU bookkeeping.
//                                                            (reference: kernels.cu:179-230, modules.py:24-118,263-366)
//   render_kernel<IMPL, RW, OBS, SHADE, NG, STEP>   (OBS = 2: pooled observations only, no plane stores in the kernel; STEP = 1: a
//                   single-agent env's physics step first, in the same wave - ms_step_render's one launch a step)
//                   one wavefront per (env, agent, 64-ray group) - or, NG = 4, per four such groups
//                   that share one list of lines, the launch's last envs left to one-group waves (256 rays and up on large
//                   launches; SHADE = 0: no shading pass, for callers that want no colour).  The lines it meets: the other agents'
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
//   explorer_kernel     the Explorer env's books between frames (reward, episode rule, forgetting) as one launch.
//                                                            (reference: demo/envs/explorer.py:45-90)
//   deathmatch_kernel   the Deathmatch env's game logic between frames (revive, crosshairs, hits and wounds, health, damage,
//                   reward, next step's dead) as one element-wise launch behind ms_render.
//                                                            (reference: demo/envs/deathmatch.py:46-88)
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
#include <atomic>
#include "../../include/megastep_hip_test.h"

namespace {

constexpr float AMBIENT = .1f;      // kernels.cu:9
constexpr float LUMINANCE = 2.f;    // kernels.cu:240
constexpr int WAVE = 64;
constexpr int WG = 256;             // 4 waves per workgroup
constexpr int WAVES = WG/WAVE;

// The library keeps NO process-wide state: what the entry points need travels in their arguments (MsConfig by value).  What
// follows is per THREAD - the last HIP error, and the test / A-B hooks of megastep_hip_test.h (ms_debug_*), which pin choices
// ms_render / ms_step_physics otherwise make from the shapes: a thread that pins one changes its own calls only, every other
// thread gets the product's behaviour.  (Round 4 had these as plain globals: a test hook changed the launches of the whole process.)
thread_local int g_last_hip_error = 0;
thread_local int g_pair_telemetry = 0;      // ms_debug_pair_telemetry
thread_local int g_ray_groups = 0;          // ms_debug_ray_groups: 0 = ms_render picks render_kernel's NG from the resolution
thread_local int g_physics_pack = 0;        // ms_debug_physics_pack: 0 = ms_step_physics picks the envs a physics wave takes side by side, k >= 1 = k
thread_local float g_tail_rounds = -1.f;    // ms_debug_ray_group_tail: < 0 = ms_render's own share of one-group waves at the end of a launch of wide ones
thread_local int g_tail_envs = -1;          //   ... >= 0: that many envs exactly
thread_local int g_last_step_fused = 0;     // ms_debug_last_step_fused: did this thread's last ms_step_render go out as one launch?
thread_local int g_last_render_groups = 0;  // ms_debug_last_render_groups: the NG this thread's last ms_render launched

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

#include "kernels/math.h"
#include "kernels/physics.h"
#include "kernels/lighting.h"
#include "kernels/render.h"
#include "kernels/bake.h"
#include "kernels/wallgrid.h"
#include "kernels/envlogic.h"

// ------------------------------------------------------------------------------------------------
// launch geometry (host): what ms_render / ms_step_physics decide before a launch - also behind ms_host_render_plan /
// ms_host_physics_pack, so that tests walk whole launches on the CPU (tests/test_launch_geometry.py)
// ------------------------------------------------------------------------------------------------
// Ray groups per wave (render_kernel's NG): an agent's groups of 64 rays share the wave's list of walls instead of each wave
// building its own.  Measured: four groups pay from 256 rays up - a quarter of the vector instructions saved
// without colour, a sixth with - IF the launch has two and a half rounds of such waves to fill the machine with and every XCD's
// last envs are left to waves of one group (render_block): 4096 x 4 x 512 rays 157.5 -> 139.0 us (colourless 127.4 -> 104.7),
// C5's share 216.7 -> 196.7; with a round or less of them - 4096 x 1 x 256 rays - a quarter is LOST.  Two groups never pay.
// `pinned`: ms_debug_ray_groups (1, 2, 4; else the rule above).  The one-group waves' share: `tail_rounds` rounds of the
// machine's wave slots' worth of the wide waves' work (< 0: half a round), or `tail_envs` envs exactly if >= 0.
struct RenderPlan { int ng; long long n_blocks; };
RenderPlan render_plan(const int n_envs, const int n_agents, const int R, const int slots, const int pinned, const float tail_rounds,
                       const int tail_envs_exact, RenderConsts& rc) {
    int ng = 1;
    if (R >= 4*WAVE && 2LL*n_envs*n_agents*((R + 4*WAVE - 1)/(4*WAVE)) >= 5LL*slots) ng = 4;   // (2.5 rounds, the one-group waves' half included)
    if (pinned == 1 || pinned == 2 || pinned == 4) ng = pinned;
    int tail = 0;
    const int G1 = (R + WAVE - 1)/WAVE;
    const int envs_lo = n_envs/8, envs_rem = n_envs % 8, envs_hi = envs_lo + (envs_rem ? 1 : 0);
    if (ng > 1) {
        const double rounds = tail_rounds >= 0.f ? (double)tail_rounds : 0.5;
        const long long tail_envs = tail_envs_exact >= 0 ? tail_envs_exact : (long long)ceil(rounds*slots*ng/((double)n_agents*G1));
        tail = (int)std::min<long long>((tail_envs + 7)/8, envs_hi);                        // per XCD
        if (tail >= envs_hi && !(pinned > 1)) ng = 1;                 // nothing left for the wide waves: the plain kernel
    }
    const int G = (R + ng*WAVE - 1)/(ng*WAVE);
    rc.by_f = divisor_of((unsigned)(n_agents*G));
    rc.by_g = divisor_of((unsigned)G);
    rc.by_f1 = divisor_of((unsigned)(n_agents*G1));
    rc.by_g1 = divisor_of((unsigned)G1);
    rc.envs_lo = envs_lo; rc.envs_rem = envs_rem; rc.tail = tail;
    // (NG > 1: every XCD as many blocks as the one with the most envs needs)
    const long long n_blocks = ng > 1 ? 8LL*((long long)(envs_hi - std::min(tail, envs_hi))*n_agents*G + (long long)std::min(tail, envs_hi)*n_agents*G1)
                                      : (long long)n_envs*n_agents*G;
    return RenderPlan{ng, n_blocks};
}

// Envs per wave (physics_kernel's PACK), with a wall grid (without one an env's walls are streamed, and there is nothing to put
// side by side) and while a wave's agents stay within half its lanes: as many as bring the launch down to about 4096 waves -
// 32768 envs of one agent are 5.3 rounds of waves with a lane or two at work each: 30.2 us; eight to a wave 9.7 (four 13.6,
// sixteen 10.2); 16384 x 4 agents 22.2 -> 13.3 (eight: 17.8), 8192 x 4 13.4 -> 10.2 - and two even at 4096 envs, where one round
// of short waves becomes half a round of waves twice as busy (8.7 -> 8.2 us; three 8.7, four 9.5; one agent per env 7.1 -> 6.4).
// `pinned`: ms_debug_physics_pack (>= 1; else the rule).
int physics_pack_of(const int n_envs, const int n_agents, const bool gridded, const int pinned) {
    int pack = 1;
    if (gridded && n_agents <= 16) {
        pack = n_envs > 6144 ? std::min((n_envs + 4095)/4096, 16) : n_envs >= 3072 ? 2 : 1;
        pack = std::max(std::min(pack, WAVE/2/n_agents), 1);
    }
    if (pinned >= 1) pack = (gridded && (long long)pinned*n_agents <= WAVE) ? pinned : 1;
    return pack;
}

// ------------------------------------------------------------------------------------------------
// host side of the C-ABI
// ------------------------------------------------------------------------------------------------
int hip_fail(hipError_t e) { g_last_hip_error = (int)e; return MS_EHIP; }

// The wave slots render_kernel has on the CURRENT device (CUs x 4 SIMDs x the waves per SIMD its registers are held to): what
// sizes a launch's choice of ray groups and its tail.  Looked up once per device (a process may drive several, of different
// sizes: round 4 kept the first caller's); a benign race - every writer stores the same number.
int wave_slots_here() {
    constexpr int MAX_DEVICES = 64;
    static std::atomic<int> slots_of[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256*4*MS_WAVES;
    if (dev >= 0 && dev < MAX_DEVICES) { const int known = slots_of[dev].load(std::memory_order_relaxed); if (known) return known; }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const int slots = cus*4*MS_WAVES_WIDE;                             // (what sizes launches of wide waves: theirs)
    if (dev >= 0 && dev < MAX_DEVICES) slots_of[dev].store(slots, std::memory_order_relaxed);
    return slots;
}

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
int ms_debug_last_render_groups(void) { return g_last_render_groups; }
int ms_debug_physics_pack(int envs) { g_physics_pack = envs; return MS_OK; }
int ms_host_physics_pack(int n_envs, int n_agents, int gridded, int pinned) { return physics_pack_of(n_envs, n_agents, gridded != 0, pinned); }
long long ms_host_render_plan(int n_envs, int n_agents, int res, int slots, int pinned_groups, float tail_rounds, int tail_envs, int* groups) {
    RenderConsts rc;
    const RenderPlan plan = render_plan(n_envs, n_agents, res, slots, pinned_groups, tail_rounds, tail_envs, rc);
    if (groups) *groups = plan.ng;
    return plan.n_blocks;
}
int ms_host_render_block(int n_envs, int n_agents, int res, int slots, int pinned_groups, float tail_rounds, int tail_envs, long long block, int* out4) {
    RenderConsts rc;
    const RenderPlan plan = render_plan(n_envs, n_agents, res, slots, pinned_groups, tail_rounds, tail_envs, rc);
    int fan = 0;
    if (block < 0 || block >= plan.n_blocks) return -1;
    return render_block((int)block, (int)plan.n_blocks, n_agents, res, plan.ng, rc, out4[0], out4[1], out4[2], out4[3], fan) ? 1 : 0;
}
int ms_debug_ray_group_tail(float rounds, int envs) { g_tail_rounds = rounds; g_tail_envs = envs; return MS_OK; }
int ms_debug_pair_telemetry(int on) { g_pair_telemetry = on ? 1 : 0; return MS_OK; }

int ms_test_arithmetic(const float* n, const float* d, float* q_inrange, float* q_ieee, const float* x, float* r_any, float* r_ieee,
                       long long count, void* stream) {
    if (count < 0 || ((q_inrange || q_ieee) && !(n && d)) || ((r_any || r_ieee) && !x)) return MS_EINVAL;
    if (count == 0) return MS_OK;
    const long long blocks = (count + WG - 1)/WG;
    if (blocks > 0x7fffffffLL) return MS_EUNSUPPORTED;
    hipLaunchKernelGGL(arithmetic_test_kernel, dim3((unsigned)blocks), dim3(WG), 0, (hipStream_t)stream, n, d, q_inrange, q_ieee, x, r_any, r_ieee, count);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MS_OK : hip_fail(e);
}

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

void ms_host_wall_sectors(float x0, float y0, float x1, float y1, const float* o, int* first, int* count, const float* w, int* sector) {
    const float cx = .5f*(x0 + x1), cy = .5f*(y0 + y1);
    wg_sectors_of(cx, cy, make_float4(o[0], o[1], o[2], o[3]), *first, *count);
    *sector = wg_sector_of(cx, cy, make_float4(w[0], w[1], w[2], w[3]));
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
        !bits_starts || !bits || ((vis_entries != nullptr) != (near_rows != nullptr)) || (!pool && !vis_entries) || (vis_entries && !sc->wg_pool_base) ||
        ((uintptr_t)near_rows % 16) || ((uintptr_t)vis_entries % 4)) return MS_EINVAL;
    if (n_reps == 0 || max_cells == 0) return MS_OK;
    const long long blocks = (2LL*max_cells + WAVES - 1)/WAVES;
    if (blocks > 0x7fffffffLL || n_reps > 65535) return MS_EUNSUPPORTED;
    hipLaunchKernelGGL(wallgrid_fill_kernel, dim3((unsigned)blocks, (unsigned)n_reps), dim3(WG), 0, (hipStream_t)stream,
                       *sc, reps, bits_starts, bits, pool, reinterpret_cast<float4*>(near_rows), vis_entries);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MS_OK : hip_fail(e);
}

// (the optional structs of ms_step_physics / ms_move_step_render, checked alike)
static bool step_options_ok(const MsMovement* mv, const MsStepExtras* ex) {
    if (mv && (!mv->actions || !mv->table || mv->n_actions < 1 || !(mv->keep == mv->keep))) return false;
    if (ex) {
        if (ex->spawn_positions && (!ex->spawn_angles || !ex->respawn_mask || !ex->respawn_choice || ex->n_spawns < 1 ||
                                    ((uintptr_t)ex->spawn_positions % 8))) return false;
        if (ex->lifespans && (!ex->max_lifespans || !ex->fresh_max)) return false;
        if (ex->imu && !(ex->imu_ang_scale == ex->imu_ang_scale && ex->imu_speed_scale == ex->imu_speed_scale)) return false;
    }
    return true;
}

int ms_step_physics(const MsScenery* sc, const MsAgents* ag, const MsMovement* mv, const MsStepExtras* ex, float* progress,
                    const MsConfig* cfg, void* stream) {
    if (!scenery_ok(sc) || !agents_ok(ag) || !progress || !config_ok(cfg)) return MS_EINVAL;
    if (!step_options_ok(mv, ex)) return MS_EINVAL;
    if (sc->wg_cells && (!sc->wg_starts || !sc->wg_geom || !sc->wg_near_rows || !(sc->wg_cell > 0.f) || ((uintptr_t)sc->wg_cells % 16) ||
                         ((uintptr_t)sc->wg_geom % 16) || ((uintptr_t)sc->wg_near_rows % 16))) return MS_EINVAL;
    const int pack = physics_pack_of(sc->n_envs, sc->n_agents, sc->wg_cells != nullptr, g_physics_pack);
    // per wave: 2 float4 + a float + an unsigned per agent, rounded up to whole float4s
    const size_t slice = ((sizeof(float)*8 + sizeof(float) + sizeof(unsigned))*(size_t)sc->n_agents*pack + 15)/16;
    if (slice*16 > 56*1024) return MS_EUNSUPPORTED;
    const MsMovement no_move{nullptr, nullptr, 0, 0.f};
    const MsStepExtras no_extras{nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, 1.f, 1.f};
    const MsMovement mvv = mv ? *mv : no_move;
    MsStepExtras exv = ex ? *ex : no_extras;
    exv.imu_ang_scale = 1.f/exv.imu_ang_scale; exv.imu_speed_scale = 1.f/exv.imu_speed_scale;   // (the kernel multiplies: see its IMU reading)
    MsScenery scn = *sc;
    if (!sc->wg_cells) { scn.wg_geom = sc->lines_vals; scn.wg_starts = sc->lines_starts; }   // (rows the kernel may read: see there)
    const hipStream_t hs = (hipStream_t)stream;
    const Divisor by_a = divisor_of((unsigned)sc->n_agents);
    // one wavefront per env (several envs per wave, one AFTER the other: 2 -> +25 %, 4 -> +85 % at 4096 envs; side by side: PACK)
#define MS_LAUNCH_PHYSICS_P(M, E, P) \
    hipLaunchKernelGGL((physics_kernel<M, E, P>), dim3((sc->n_envs + pack - 1)/pack), dim3(WAVE), slice*16, hs, scn, *ag, progress, cfg->agent_radius, cfg->fps, mvv, exv, pack, by_a)
#define MS_LAUNCH_PHYSICS(M, E) { if (pack > 1) MS_LAUNCH_PHYSICS_P(M, E, 1); else MS_LAUNCH_PHYSICS_P(M, E, 0); }
    if (mv && ex) MS_LAUNCH_PHYSICS(1, 1)
    else if (ex) MS_LAUNCH_PHYSICS(0, 1)
    else if (mv) MS_LAUNCH_PHYSICS(1, 0)
    else MS_LAUNCH_PHYSICS(0, 0)
#undef MS_LAUNCH_PHYSICS_P
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

// ms_render, and - with `progress` - ms_step_render's fused launch (returns MS_EUNSUPPORTED, having launched nothing, where that
// one does not apply: the caller then makes the two calls)
static int render_launch(const MsScenery* sc, const MsAgents* ag, const MsRender* out, const MsConfig* cfg, void* stream, float* progress,
                         const MsMovement* mv = nullptr, const MsStepExtras* ex = nullptr) {
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
    const int slots = wave_slots_here();                                 // the current device's wave slots for this kernel
    bool wide_ok = true;
#if MS_AB_IMPLS
    if (getenv("MEGASTEP_RENDER_IMPL")) wide_ok = false;
#endif
    // (without a light grid the rays that land on an agent are lit by dynlight_kernel, which takes them by groups of 64)
    if (!(sc->lg_vals && sc->lg_starts && sc->lg_geom && sc->lg_cell > 0.f) && sc->n_agents > 1 && (out->screen || out->obs_rgb)) wide_ok = false;
    RenderConsts rc;
    const RenderPlan plan = render_plan(sc->n_envs, sc->n_agents, R, slots, wide_ok ? g_ray_groups : 1, g_tail_rounds, g_tail_envs, rc);
    const int ng = plan.ng;
    g_last_render_groups = ng;
    const long long n_fans = plan.n_blocks;
    // the workspace's layout (MS_RENDER_WORKSPACE_INTS): 16 counters, a queue of one entry per (env, agent, 64 rays), the headings
    const long long ws_queue = (long long)sc->n_envs*sc->n_agents*((R + WAVE - 1)/WAVE);
    if (n_fans > 0x7fffffffLL || ws_queue > 0x7fffff00LL) return MS_EUNSUPPORTED;
    rc.ws_headings = 16 + (int)((ws_queue + 1) & ~1LL);
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
    const bool walls_listed = sc->wg_cells && sc->wg_starts && sc->wg_geom && sc->wg_pool && sc->wg_pool_base && sc->wg_cell > 0.f &&
                              cfg->agent_radius*1.001f < sc->wg_near && 1.f + half_screen*half_screen <= WG_MAX_RU2;
    if (walls_listed && (((uintptr_t)sc->wg_cells % 16) || ((uintptr_t)sc->wg_geom % 16))) return MS_EINVAL;
    if (!walls_listed) {                                                 // (the kernel reads a row of each whatever happens: see there)
        scn.wg_cells = nullptr;
        scn.wg_geom = sc->lines_vals;                                    // at least 16 bytes per env: every env has its agents' lines
        scn.wg_starts = sc->lines_starts;
        scn.wg_pool_base = reinterpret_cast<const long long*>(sc->lines_vals);   // (16 bytes of lines per env at least: 8 are there)
    }
    // Headings: from ms_physics' cache when the agents carry one and a single kernel does the whole job (then the
    // workspace is not needed at all); otherwise from render_prep_kernel, which also resets the workspace's counters.
    MsAgents agn = *ag;
    MsRender outn = *out;
    if (out->obs_depth) outn.obs_max_depth = 1.f/out->obs_max_depth;     // (the kernel multiplies: see the pooled depth in render.h)
    const bool colour = out->screen || out->obs_rgb;                      // else: render_kernel<.,.,1,0>, which has no pass 3
    const bool one_kernel = grid || sc->n_agents == 1 || !colour;        // (nothing to light without colour)
    if (ag->headings && one_kernel) {
        if ((uintptr_t)ag->headings % 16) return MS_EINVAL;
        outn.workspace = nullptr;
    } else if (!progress) {
        agn.headings = nullptr;
        if (out->workspace) {
            if ((uintptr_t)out->workspace % 8) return MS_EINVAL;
            const int na = sc->n_envs*sc->n_agents;
            hipLaunchKernelGGL(render_prep_kernel, dim3((na + WG - 1)/WG), dim3(WG), 0, (hipStream_t)stream,
                               *ag, out->workspace, na, rc.ws_headings);
        }
    }
    // dynlight_kernel reads the per-ray outputs back: only the one-kernel path can do without some of them
    const bool all_planes = out->indices && out->locations && out->dots && out->distances && out->screen;
    const bool pooled = out->obs_rgb || out->obs_depth || out->obs_centre || out->seen_stamp;
    if (colour && (!all_planes || pooled) && !(grid || sc->n_agents == 1)) return MS_EUNSUPPORTED;   // dynlight_kernel patches `screen` afterwards
    if (!all_planes && !pooled && !out->indices && !out->locations && !out->dots && !out->distances && !out->screen) return MS_EINVAL;
    const bool obs = pooled || !all_planes;
    if (progress) {
        // The fused step (render_kernel<..., STEP = 1>): one agent per env and at most 64 rays - the agent is ONE wave, which
        // runs the env's physics first and renders from the pose it ends on; the product raycast; and a wall grid that either
        // serves both halves of the step or neither (ms_render goes without it when the call's near plane or field of view is
        // outside what its vis lists were built for, ms_step_physics never does).
        bool older = false;
#if MS_AB_IMPLS
        older = seq || pairs1;
#endif
        const bool physics_listed = sc->wg_cells != nullptr;
        if (sc->n_agents != 1 || R > WAVE || ng != 1 || older || physics_listed != walls_listed ||
            (physics_listed && (!sc->wg_near_rows || ((uintptr_t)sc->wg_near_rows % 16)))) return MS_EUNSUPPORTED;
        agn = *ag;                                                       // (the wave works the heading out itself and leaves it in the cache, if there is one)
        outn.workspace = nullptr;
    }
    rc.x_clip = 0.5f*cfg->agent_radius/sqrtf(1.f + half_screen*half_screen);
    rc.c_b = 0.5f*(float)R/half_screen;
    rc.by_m = divisor_of((unsigned)sc->n_model);
    rc.skip_own = (sc->model_radius > 0.f && sc->model_radius*1.01f < cfg->agent_radius) ? 1 : 0;
    rc.inv_res = ((R & (R - 1)) == 0 && half_screen > 1e-3f) ? 1.f/(float)R : 0.f;
    rc.telemetry = g_pair_telemetry;
    bool older_raycast = false;
#if MS_AB_IMPLS
    older_raycast = seq || pairs1;                                       // (their instantiations are the colour ones, whatever is asked for)
#endif
    // (the instantiations with optional outputs - every colourless one, and with MS_OBS_MASK the colour one of pooled observations
    // at one ray group a wave - read which are wanted from here: see OUT_* in render.h)
    if ((!colour || (MS_OBS_MASK && obs && ng == 1)) && !older_raycast)
        outn.obs_subsample = (out->obs_subsample & 0xff) | (((out->indices ? OUT_INDICES : 0) | (out->locations ? OUT_LOCATIONS : 0) |
                              (out->dots ? OUT_DOTS : 0) | (out->distances ? OUT_DISTANCES : 0) | (out->obs_depth ? OUT_DEPTH : 0) |
                              (out->obs_centre ? OUT_CENTRE : 0) | (out->seen_stamp ? OUT_SEEN : 0) | (out->screen ? OUT_SCREEN : 0) |
                              (out->obs_rgb ? OUT_RGB : 0)) << 8);
    constexpr int RW = 1;
    const int rblocks = (int)((n_fans + RW - 1)/RW);
    const dim3 rgrid(rblocks), rblock(RW*WAVE);
    const hipStream_t hs = (hipStream_t)stream;
    // (colour, and not one per-ray plane wanted - the demo envs' request: the instantiation that has no plane stores in it)
    [[maybe_unused]] const bool no_planes = colour && !out->indices && !out->locations && !out->dots && !out->distances && !out->screen;
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
    { if (!colour) MS_LAUNCH_RENDER_NG(1, 0, NG_); else if (no_planes) MS_LAUNCH_RENDER_NG(2, 1, NG_); else if (obs) MS_LAUNCH_RENDER_NG(1, 1, NG_); \
      else MS_LAUNCH_RENDER_NG(0, 1, NG_); }
    if (progress) {
        RenderConstsStep rcs;
        static_cast<RenderConsts&>(rcs) = rc;
        rcs.progress = progress; rcs.fps = cfg->fps; rcs.wg_cells_physics = sc->wg_cells;
        rcs.mv = mv ? *mv : MsMovement{nullptr, nullptr, 0, 0.f};
        rcs.ex = ex ? *ex : MsStepExtras{nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, 1.f, 1.f};
        rcs.ex.imu_ang_scale = 1.f/rcs.ex.imu_ang_scale; rcs.ex.imu_speed_scale = 1.f/rcs.ex.imu_speed_scale;
#define MS_LAUNCH_STEP(O, S) \
    hipLaunchKernelGGL((render_kernel<2, RW, O, S, 1, 1>), rgrid, rblock, 0, hs, scn, agn, outn, cfg->agent_radius, half_screen, R, (int)n_fans, rcs)
        if (!colour) MS_LAUNCH_STEP(1, 0); else if (obs) MS_LAUNCH_STEP(1, 1); else MS_LAUNCH_STEP(0, 1);
#undef MS_LAUNCH_STEP
    }
    else if (ng == 4) MS_LAUNCH_RENDER_OS(4)
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

int ms_deathmatch_shoot(int n_envs, int n_agents, const MsDeathmatch* dm, void* stream) {
    if (n_envs <= 0 || n_agents <= 0 || !dm || !dm->centre || !dm->positions || !dm->upper || !dm->health || !dm->damage || !dm->dead ||
        ((uintptr_t)dm->centre % 8) || ((uintptr_t)dm->positions % 8) || ((uintptr_t)dm->upper % 8) ||
        !(dm->clearance == dm->clearance) || !(dm->hit_damage == dm->hit_damage) || !(dm->tick_damage == dm->tick_damage)) return MS_EINVAL;
    const long long rows = (long long)n_envs*n_agents, blocks = (rows + WG - 1)/WG;
    if (blocks > 0x7fffffffLL) return MS_EUNSUPPORTED;
    hipLaunchKernelGGL(deathmatch_kernel, dim3((unsigned)blocks), dim3(WG), 0, (hipStream_t)stream, *dm, n_envs, n_agents);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MS_OK : hip_fail(e);
}

int ms_explorer_books(int n_envs, const MsExplorer* ex, void* stream) {
    if (n_envs <= 0 || !ex || !ex->tally || !ex->before || !ex->lengths || !ex->epoch || !ex->over || !ex->reward || ex->pixels <= 0 ||
        ((uintptr_t)ex->tally % 4) || ((uintptr_t)ex->before % 4) || ((uintptr_t)ex->lengths % 4) || ((uintptr_t)ex->epoch % 4)) return MS_EINVAL;
    hipLaunchKernelGGL(explorer_kernel, dim3((unsigned)((n_envs + WG - 1)/WG)), dim3(WG), 0, (hipStream_t)stream, *ex, n_envs);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MS_OK : hip_fail(e);
}

int ms_render(const MsScenery* sc, const MsAgents* ag, const MsRender* out, const MsConfig* cfg, void* stream) {
    return render_launch(sc, ag, out, cfg, stream, nullptr);
}

int ms_move_step_render(const MsScenery* sc, const MsAgents* ag, const MsMovement* mv, const MsStepExtras* ex, float* progress,
                        const MsRender* out, const MsConfig* cfg, void* stream) {
    if (!progress || !step_options_ok(mv, ex)) return MS_EINVAL;
    const int fused = render_launch(sc, ag, out, cfg, stream, progress, mv, ex);
    g_last_step_fused = fused == MS_OK ? 1 : 0;
    if (fused != MS_EUNSUPPORTED) return fused;
    const int p = ms_step_physics(sc, ag, mv, ex, progress, cfg, stream);
    return p != MS_OK ? p : render_launch(sc, ag, out, cfg, stream, nullptr);
}
int ms_step_render(const MsScenery* sc, const MsAgents* ag, float* progress, const MsRender* out, const MsConfig* cfg, void* stream) {
    return ms_move_step_render(sc, ag, nullptr, nullptr, progress, out, cfg, stream);
}
int ms_debug_last_step_fused(void) { return g_last_step_fused; }

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
