/* megastep_hip_test.h -- what libmegastep_hip.so exports for its own test suite and A/B runs, on top of megastep_hip.h:
 * host instantiations of the device functions the kernels cull with (so that the CPU suite can check every cull against
 * the oracle without a GPU), and debug switches.  Nothing here is part of the drop-in boundary. */
#ifndef MEGASTEP_HIP_TEST_H
#define MEGASTEP_HIP_TEST_H
#include "megastep_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Host instantiation of render_kernel's pass 1 for one line (reference: the all-lines loop kernels.cu:352-377, which it
 * culls): pose = (x, y, sin, cos of the heading), line = (ax, ay, bx, by), `group` = which 64 rays of the agent's `res`:
 * rays first .. first + count - 1 of the group (0-based within it) are the only ones the kernel intersects with the line. */
void ms_host_ray_interval(const float* pose, const float* line, int res, float fov, float agent_radius, int group, int* first, int* count);
/* ... and for a wave that serves `groups` (1, 2, 4) groups of 64 rays (render_kernel's NG): `wave` = which run of 64 x groups
 * rays of the agent's; first / count within that run. */
void ms_host_ray_interval_wide(const float* pose, const float* line, int res, float fov, float agent_radius, int groups, int wave,
                               int* first, int* count);
/* Pins the number of 64-ray groups a render wave serves (1, 2, 4; 0 = ms_render picks it from the request: DESIGN 3.6).
 * Per calling THREAD (like every ms_debug_* switch: the library keeps no process-wide state - a thread that pins something
 * changes its own calls only); for A/B runs and tests - every setting produces the same bits. */
int ms_debug_ray_groups(int groups);
/* Pins the number of envs a physics wave takes side by side (physics_kernel's PACK; 0 = ms_step_physics picks it from the
 * world's size, 1 = one, k = k where k x n_agents <= 64 and there is a wall grid, else one).  Per calling thread; A/B runs and
 * tests - every setting produces the same bits. */
int ms_debug_physics_pack(int envs);
/* What the calling thread's last ms_render launched: the 64-ray groups per wave (render_kernel's NG: 1, 2, 4) it settled on -
 * the pins, the scenery (no light grid: one group), the build (A/B raycasts: one) and the device's wave slots all taken into
 * account, which a caller re-deriving the rule cannot know; 0 before the thread's first call.  bench.py labels its
 * lines with it. */
int ms_debug_last_render_groups(void);
/* Did the calling thread's last ms_step_render go out as one launch (1) or as ms_physics + ms_render (0)? */
int ms_debug_last_step_fused(void);
/* The launch geometry ms_render / ms_step_physics decide on the host, and the render kernel's own block -> rays mapping
 * (render_block), for tests that walk whole launches on the CPU.  `slots`: the machine's wave slots for the render kernel (CUs x
 * 4 SIMDs x 6 waves; 6144 on MI355X); pinned_groups / tail_rounds / tail_envs as the ms_debug_* hooks (0 / < 0 / < 0: the rules).
 * ms_host_render_plan returns the number of one-wave blocks and the ray groups per wave; ms_host_render_block fills out4 =
 * (env, agent, first ray, rays) for block `block` and returns 1, or 0 for a spare block, -1 outside the launch.
 * ms_host_physics_pack: the envs a physics wave takes side by side. */
long long ms_host_render_plan(int n_envs, int n_agents, int res, int slots, int pinned_groups, float tail_rounds, int tail_envs, int* groups);
int ms_host_render_block(int n_envs, int n_agents, int res, int slots, int pinned_groups, float tail_rounds, int tail_envs, long long block, int* out4);
int ms_host_physics_pack(int n_envs, int n_agents, int gridded, int pinned);
/* A launch of waves of several groups ends with waves of one group for its last envs; their share, in rounds of the machine's
 * wave slots' worth of the wide waves' work (< 0: ms_render's own, half a round; 0: none; large: every env), or, if
 * envs >= 0, that many envs exactly.  Per calling thread; A/B runs and tests - every setting produces the same bits. */
int ms_debug_ray_group_tail(float rounds, int envs);
/* Has ms_render's waves add their (line, ray) pair and pair-window counts to workspace[3] and [4] (two atomics per wave on
 * one address: milliseconds at 10^5 waves - tools/pair_stats.py only).  Per calling thread. */
int ms_debug_pair_telemetry(int on);
/* The kernels' arithmetic shortcuts against what they stand for, element by element on the device (DEVICE pointers, `count`
 * elements each): q_inrange[i] = div_inrange(n[i], d[i]) - the division without range scaling the render kernel uses where its
 * operands are in range by construction (kernels/math.h) - next to q_ieee[i] = n[i] / d[i] as the compiler expands a correctly
 * rounded division; r_any[i] = sqrt_any(x[i]) next to r_ieee[i] = sqrtf(x[i]).  tests/test_gpu_numerics.py holds them against
 * each other bit for bit over the ranges the call sites guarantee.  Any output pointer may be NULL. */
int ms_test_arithmetic(const float* n, const float* d, float* q_inrange, float* q_ieee, const float* x, float* r_any, float* r_ieee,
                       long long count, void* hip_stream);
/* Host instantiation of the light grid's build (ms_bake; accelerates kernels.cu:238-268) for one cell c (row-major in a grid
 * of nx x ny cells of size `cell` from (ox, oy)), over n_walls walls (n_walls x 4 floats: ax, ay, bx, by) and n_lights
 * lights (n_lights x 3: x, y, intensity), HOST memory: words[4] = the lights' 2-bit verdicts as in lg_vals (0 unknown, 1 lit,
 * 2 dark); candidates = the (light, wall) pairs of the cell's list as in lg_pool (0x80000000 | light << 24 | wall), at most
 * max_candidates of them written; returns how many there are. */
int ms_host_lightgrid_cell(const float* walls, int n_walls, const float* lights, int n_lights, float ox, float oy, int nx, int ny,
                           float cell, int c, unsigned* words, unsigned* candidates, int max_candidates);
/* Host restatement of how render_kernel's pass 2 settles a ray's nearest hit (reference: the order-dependent fold
 * kernels.cu:369-376): the n_hits hits (s[i] > near plane, line[i]) go through the kernel's three key slots in the order
 * `order` (a permutation of 0..n_hits-1), 64 to a window, lockstep within a window as a wavefront plays them.  Returns 1
 * when the slots cannot tell (the kernel then redoes the ray by the literal fold), else 0 with the hit in *nearest_s /
 * *nearest_line (-1: none). */
int ms_host_fold_hits(const float* s, const int* line, int n_hits, const int* order, float* nearest_s, int* nearest_line);
/* Host instantiation of ms_physics' reach cull in front of the agent-agent collision test (reference: kernels.cu:119-133,
 * 193-200), for CPU tests: me, other = (x, y, vx/fps, vy/fps); 1 = the pair cannot collide this step, the test is skipped. */
int ms_host_agents_apart(const float* me, const float* other, float agent_radius);
/* ... and of the reach cull in front of the agent-wall test (kernels.cu:135-171,202-205): agent = (x, y, vx/fps, vy/fps),
 * wall = (ax, ay, bx, by); 1 = the wall is beyond the agent's reach this step, the test is skipped.  (The host evaluates the
 * foot of the perpendicular with a true division, the kernel with v_rcp_f32: both are lower bounds on the distance.) */
int ms_host_wall_beyond_reach(const float* agent, const float* wall, float agent_radius);
/* ... and the reach itself, which also picks the tier of the cell's near list (wg_reach_lo, wg_reach) or the sweep. */
float ms_host_wall_reach(const float* agent, float agent_radius);
/* Host instantiation of the scan's test, for CPU tests: does wall o = (ax, ay, bx, by) hide wall w from every point of
 * the cell [x0, x1] x [y0, y1] (as ms_wallgrid_scan grows it) for near planes below `near_plane`? */
int ms_host_wall_hidden(float x0, float y0, float x1, float y1, const float* o, const float* w, float near_plane);
/* The scan's sort of a cell's occluders into sectors of directions (wallgrid_scan_kernel, WG_SECTORS): the run of sectors
 * [first, first + count) modulo 64 that occluder o touches as seen from the centre of the cell [x0, x1] x [y0, y1], and the one
 * sector target w's middle lies in - a target is only tried against the occluders of its sector, so whenever ms_host_wall_hidden
 * says o hides w from the cell, `sector` must be in o's run (tests/test_wallgrid.py). */
void ms_host_wall_sectors(float x0, float y0, float x1, float y1, const float* o, int* first, int* count, const float* w, int* sector);
/* ... and of the scan of one whole cell (c, row-major in a grid of nx x ny cells of size `cell` from (ox, oy)) over
 * n_walls walls (n_walls x 4 floats, HOST memory): vis[t] = 1 where wall t goes on the cell's vis list, close[t] = 2 / 1
 * where it is within reach_lo / reach of the cell. */
void ms_host_wallgrid_cell(const float* walls, int n_walls, float ox, float oy, int nx, int ny, float cell, int c,
                           float near_plane, float reach_lo, float reach, unsigned char* vis, unsigned char* close);

/* ... and of the arcs: the run of steps [*lo8, *hi8] (modulo 256) of directions wall w can be seen in from the cell; and
 * whether a wave whose rightmost / leftmost ray point along (right_x, right_y) / (left_x, left_y) would keep such a wall. */
void ms_host_wall_arc(float x0, float y0, float x1, float y1, const float* w, int* lo8, int* hi8);
int  ms_host_wedge_meets(float right_x, float right_y, float left_x, float left_y, int lo8, int hi8);

/* Scalar helper exported for tests: sin(pi x), cos(pi x) exactly as the kernels evaluate them. */
void ms_host_sincospi(float x, float* s, float* c);
/* Helpers exported for tests of ms_bake's culling (host instantiations of the device functions): the angular bin
 * (of MS_BAKE_BINS around a light) a point falls in, and the circular run of bins [first, first + count) a wall can
 * shadow.  A wall obstructs a point from the light (kernels.cu:238-259) only if the point's bin is in the wall's run. */
#define MS_BAKE_BINS 64
int  ms_host_bake_point_bin(float light_x, float light_y, float x, float y);   /* -1: undecidable, test every wall */
void ms_host_bake_wall_bins(float light_x, float light_y, float ax, float ay, float bx, float by, int* first, int* count);

#ifdef __cplusplus
}
#endif
#endif /* MEGASTEP_HIP_TEST_H */
