/*
 * megastep_hip.h -- C-ABI of the MI355X (gfx950) simulation core.
 *
 * This is the drop-in boundary for the reference's native extension `megastepcuda`
 * (surfaced in Python as `megastep.cuda`, /root/reference/megastep/__init__.py:7-20,
 * bound in /root/reference/megastep/src/wrappers.cpp:30-173).  Every entry point below
 * names the reference interface it replaces.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers + sizes, no torch / ATen types.
 *   - the library never allocates, frees or synchronises: the caller owns every buffer
 *     and all work is enqueued asynchronously on the hipStream_t it passes
 *     (the reference enqueues on at::cuda::getCurrentCUDAStream(), kernels.cu:30-32).
 *   - configuration travels by value with each call (the reference keeps it in
 *     process-global __constant__ memory, kernels.cu:12-27), so one process can drive
 *     several devices / configurations.
 *   - every function returns MS_OK (0) or a negative MS_E* code; ms_strerror() explains it.
 *     (The reference raises c10::Error through pybind, common.h:12-14,33-37.)
 *   - all float data is IEEE binary32, all index data int32, as in the reference
 *     (rebar/arrdict.py:79-88, common.h:122).
 */
#ifndef MEGASTEP_HIP_H
#define MEGASTEP_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MS_ABI_VERSION 16

#define MS_OK            0
#define MS_EINVAL       -1   /* bad argument (null pointer, non-positive size, ...) */
#define MS_EHIP         -2   /* a HIP runtime call failed; see ms_last_hip_error()  */
#define MS_EUNSUPPORTED -3   /* shape outside what the kernels support              */
#define MS_ENODEVICE    -4   /* no usable gfx950 device                             */

/* Replaces `initialize(agent_radius, res, fov, fps)` (wrappers.cpp:53, kernels.cu:18-27). */
typedef struct MsConfig {
    float agent_radius;   /* collision radius and near plane, metres (core.py:14)  */
    int   res;            /* rays per agent, R                                      */
    float fov;            /* field of view, degrees (< 180)                         */
    float fps;            /* simulation steps per second                            */
} MsConfig;

/* Replaces `Scenery` + its three `Ragged`s (common.h:102-155,179-214). Device pointers. */
typedef struct MsScenery {
    int n_envs, n_agents, n_model;   /* N, A, M = model.size(0)                                */
    const float* lights_vals;        /* (sum I, 3)  x, y, intensity                            */
    const int*   lights_widths;      /* (N,)                                                   */
    const int*   lights_starts;      /* (N,)                                                   */
    float*       lines_vals;         /* (sum L, 2, 2) [endpoint][xy]; rows [0, A*M) of each env
                                        are the agents' models and are REWRITTEN by ms_render
                                        (kernels.cu:316-317). Must be 16-byte aligned.         */
    const int*   lines_widths;       /* (N,)                                                   */
    const int*   lines_starts;       /* (N,)                                                   */
    const int*   lines_inverse;      /* (sum L,) global line -> env                            */
    const float* textures_vals;      /* (sum T, 3) linear RGB, ragged per GLOBAL line          */
    const int*   textures_widths;    /* (sum L,) texels per line                               */
    const int*   textures_starts;    /* (sum L,)                                               */
    const int*   textures_inverse;   /* (sum T,) texel -> global line                          */
    const float* model;              /* (M, 2, 2) agent outline in the agent frame             */
    float*       baked_vals;         /* (sum T,) written by ms_bake, read by ms_render         */
    int n_lines_total, n_lights_total, n_texels_total;
    /* Optional light grid (lg_vals NULL = none; no counterpart in the reference): a per-env uniform grid over the
     * floorplan that caches, for every cell and each of the env's first 64 lights, whether the light reaches the
     * cell.  Written by ms_bake, read by ms_render's dynamic lighting, exact by construction: a cell is only ever
     * marked when EVERY point in it provably has that status; anything else stays 0 (unknown) and is worked out
     * per ray - against the cell's candidate list (the walls that could not be ruled out for the cell and that
     * light) when it has one, against every wall otherwise.
     *   lg_vals   (sum cells, 4) uint32, 2 bits per light: 0 unknown, 1 lit, 2 dark; must start zeroed
     *   lg_starts (N,)   first cell of env n
     *   lg_geom   (N, 4) float: x and y of the grid's origin, cells along x, cells along y
     *   lg_cell   cell size in metres;  lg_max_cells  max over envs of cells (launch bound for ms_bake)
     *   lg_list   (sum cells, 2) uint32, optional (NULL together with lg_pool), must start zeroed:
     *             [first pool word of the cell's candidates, 0x80000000 | how many]; second word 0 = no list
     *   lg_pool   (lg_pool_size,) uint32: word 0 is ms_bake's allocation cursor, then candidates
     *             0x80000000 | light << 24 | wall (index among the env's static lines).  A pool that runs out
     *             only costs speed: the cells that did not fit go without a list.
     * Only for sceneries with at most 64 lights in every env: pass lg_vals = NULL otherwise. */
    unsigned*    lg_vals;
    const int*   lg_starts;
    const float* lg_geom;
    float        lg_cell;
    int          lg_max_cells;
    unsigned*    lg_list;
    unsigned*    lg_pool;
    int          lg_pool_size;
    /* Optional, with lg_pool: (lg_pool_size, 4) float32 - next to every candidate of lg_pool its wall's row as the shadow
     * test wants it, (ax, ay, bx - ax, by - ay), written by ms_bake.  ms_render then has a list's walls in the trip that
     * brings its candidates, instead of one trip later (the rays that need them are the ones a launch waits for).  NULL:
     * ms_render fetches the rows from `lines` by the candidates' wall numbers. */
    float*       lg_pool_rows;
    /* Optional sharing of static geometry between envs (NULL = every env on its own; no counterpart in the reference,
     * whose scene.py:75-100 and kernels.cu:270-293 redo identical floorplans env by env): env_geom[n] is the FIRST env
     * whose walls and light POSITIONS are bit-identical to env n's (itself for a representative).  Light intensities
     * and textures stay per env.  ms_bake then works out light visibility - the O(texels x lights x walls) part,
     * which depends on walls and light positions only - once per representative, and members of a group may share one
     * light grid (equal lg_starts / lg_geom rows); the per-env part (the sum over unblocked lights with the env's
     * own intensities, kernels.cu:261-267) runs for every env as before, so baked_vals are the reference's bit for bit. */
    const int*   env_geom;
    /* Optional scratch for ms_bake (NULL = its self-contained one-pass kernel): one visibility bit per
     * (texel, light) of every representative env.  The bits of env n's representative start at word
     * bake_vis_starts[n] and take lights(n) x ceil(texels(n)/64) 64-bit words, row per light.  Contents undefined
     * before and after the call. */
    unsigned long long* bake_vis;
    const long long*    bake_vis_starts;   /* (N,) */
    long long           bake_vis_words;    /* size of bake_vis, for bounds checking */
    /* Optional wall grid (wg_cells NULL = none; no counterpart in the reference, whose kernels meet every line of an
     * env for every ray and every agent, kernels.cu:203-205,352-377): a per-floorplan uniform grid whose cells each
     * hold two lists of static walls (indices among the env's static lines, i.e. line index - n_agents*n_model):
     *   vis   every wall that can matter to a ray cast from ANY point of the cell: a wall is left out only when one
     *         other wall provably stands between it and the whole cell - with room to spare for the reference's
     *         1e-4 hysteresis, its near plane and its rounding - so that the order-dependent nearest-hit fold over
     *         the listed walls ends exactly where the fold over all of them does;
     *   near  every wall that comes within wg_reach of the cell: all an agent in the cell whose step reaches no
     *         farther than that can collide with.
     * ms_render / ms_physics look the agent's cell up and walk its list instead of the env's lines; agents outside
     * the grid, or faster than wg_reach allows, meet every line as before.  Filled by ms_wallgrid_scan +
     * ms_wallgrid_fill (below) from the static walls as they are at that moment: the walls must not move afterwards
     * (or the grid must be rebuilt / dropped).  Envs that share their walls (env_geom) share their cells.
     *   wg_cells  (sum cells + 1, 4) uint32: [first vis entry (in wg_pool, from wg_pool_base[n]), vis count, first near entry (in
     *             wg_near_rows), near count within wg_reach_lo | near count in all << 16]
     *   wg_starts (N,) first cell of env n;   wg_geom (N, 4) float: grid origin x, y, cells along x, cells along y
     *             (0 cells: this env has no grid);   wg_cell: cell size in metres
     *   wg_pool_base (N,) int64, required with wg_cells: where in wg_pool the vis lists of env n's floorplan start, in entries;
     *             a cell's "first vis entry" counts from there.  (Round 5: a world of 4096 distinct 1000-wall floorplans - the
     *             reference's cubicasa pool is 4492 - has 6 x 10^9 vis entries at 0.25 m cells, more than a 32-bit offset
     *             reaches; per floorplan it is a few million.)  Envs that share their walls share the value.
     *   wg_pool   the vis lists, one uint32 per entry: wall index | first step << 16 | last step << 24 of the arc of
     *             directions the wall can be seen in from the cell (steps of 1/64 of a quarter turn-like unit, modulo 256:
     *             ms_render skips entries whose arc misses its rays'); at least 64 entries longer than the lists need
     *   wg_near_rows  (., 4) float: the near lists as copies of the walls' rows (ax, ay, bx, by), per cell the walls
     *             within wg_reach_lo of it first, then those within wg_reach
     *   wg_near   vis lists hold for near planes (MsConfig.agent_radius) below this and fields of view up to
     *             MS_WALLGRID_MAX_FOV degrees; ms_render ignores the grid otherwise */
    const unsigned*       wg_cells;
    const int*            wg_starts;
    const float*          wg_geom;
    float                 wg_cell;
    float                 wg_reach_lo, wg_reach;
    float                 wg_near;
    const unsigned*       wg_pool;
    const long long*      wg_pool_base;
    const float*          wg_near_rows;
    /* Optional: the largest distance of a point of `model` from the agent's origin, 0 = not known.  When it is below
     * the near plane (MsConfig.agent_radius, as in the reference: core.py:14, scene.py:25-33), no ray of an agent can
     * hit the agent's own outline (kernels.cu:369) and ms_render does not try. */
    float model_radius;
} MsScenery;

/* Replaces `Agents` (common.h:157-177). Updated IN PLACE by ms_physics. */
typedef struct MsAgents {
    float* angles;        /* (N, A)    degrees           */
    float* positions;     /* (N, A, 2) metres            */
    float* angvelocity;   /* (N, A)    degrees / second  */
    float* velocity;      /* (N, A, 2) metres / second   */
    /* Optional heading cache (NULL = none; no counterpart in the reference), (N, A, 4): [angle, sin, cos, unused].
     * ms_physics, which has each agent's new angle in hand, leaves its sine and cosine here; ms_render uses an
     * entry when its angle equals the agent's current one bit for bit, and works the pair out itself otherwise
     * (an agent turned by the caller in between).  Same function, same bits either way - it saves ms_render a launch.
     * Must start as NaNs (or any value no angle takes); pass NULL to ms_render to get its self-contained path. */
    float* headings;
} MsAgents;

/* Replaces `Render` (common.h:216-222). Caller-allocated outputs.  With a light grid in the scenery any of the
 * five per-ray outputs may be NULL (not wanted: skipped); without one all five are required. */
typedef struct MsRender {
    int*   indices;       /* (N, A, R)    line index within the env, -1 on a miss */
    float* locations;     /* (N, A, R)    position along the line, NaN on a miss  */
    float* dots;          /* (N, A, R)    ray . line direction,    NaN on a miss  */
    float* distances;     /* (N, A, R)    metres, +inf on a miss                  */
    float* screen;        /* (N, A, R, 3) linear RGB, 0 on a miss                 */
    /* Optional scratch, not an output: lets ms_render compute every agent's sin/cos once, ahead of the
     * raycast, and hand the ray groups that need dynamic lighting from its first kernel to its second as
     * a compact list.  At least MS_RENDER_WORKSPACE_INTS(N, A, R) 4-byte words, 8-byte aligned, contents
     * undefined before and after the call; NULL selects slower in-kernel paths. */
    int*   workspace;
    /* Optional pooled observations, written by the render kernel itself instead of by a chain of host-side tensor
     * ops over the per-ray outputs (replaces modules.py:138-145,170-184,211-224: downsample().mean(), Depth, RGB).
     * Each pixel is the mean over `obs_subsample` adjacent rays; obs_subsample must be a power of two dividing
     * both 64 and the resolution.  NULL = not wanted.
     *   obs_rgb    (N, A, 3, R/obs_subsample)  channel-major, as the reference's RGB module returns it
     *   obs_depth  (N, A, R/obs_subsample)     mean of 1 - clamp((distance - agent_radius)/obs_max_depth, 0, 1), the quotient taken
     *                                          as ATen takes a tensor over a scalar: times the binary32 reciprocal */
    float* obs_rgb;
    float* obs_depth;
    int    obs_subsample;
    float  obs_max_depth;
    /* Optional, with obs_subsample set: what the two central observation pixels of each agent show - the index of
     * the agent whose outline the pixel's middle ray (ray pixel*obs_subsample + obs_subsample/2) landed on, else -1.
     * It is all Deathmatch reads from `indices` (demo/envs/deathmatch.py:54-58,74-80).  (N, A, 2) */
    int*   obs_centre;
    /* Optional first-sight bookkeeping (replaces demo/envs/explorer.py:34-58, a scatter over every texel of every env
     * per step): each ray that hits marks the texel under it - textures_starts[line] + min(floor(width*location),
     * width - 1), explorer.py:38-41 - by writing its env's epoch into seen_stamp, and texels whose stamp was not the
     * epoch yet are counted into seen_count.  Bumping an env's epoch forgets all its texels at once.
     *   seen_stamp (sum T,) int, seen_epoch (N,) int, seen_count (N,) int (added to, atomically) */
    int*       seen_stamp;
    const int* seen_epoch;
    int*       seen_count;
} MsRender;

/* 4-byte words of MsRender.workspace needed for N envs, A agents, R rays */
#define MS_RENDER_WORKSPACE_INTS(N, A, R) (18 + (long long)(N)*(A)*(((R) + 63)/64) + 2*(long long)(N)*(A))

int         ms_abi_version(void);
const char* ms_strerror(int code);
/* hipError_t of the most recent failing HIP call made by this library on this thread (0 if none). */
int         ms_last_hip_error(void);
/* Number of visible HIP devices, or MS_ENODEVICE. Lets the host side fail loudly up front. */
int         ms_device_count(void);

/* Replaces `bake(scenery)` (wrappers.cpp:61, kernels.cu:270-293): static lighting of every texel
 * into scenery->baked_vals. `config` is unused by this stage and may be NULL (the reference's bake reads none of the
 * initialize() constants and runs before any Core exists, scene.py:98). */
int ms_bake(const MsScenery* scenery, const MsConfig* config, void* hip_stream);

/* Optional prologue of the physics step: what the reference's movement modules do with a handful of tensor ops right
 * before they call physics (modules.py:24-66 SimpleMovement, :68-118 MomentumMovement).  Per agent, with (dx, dy, dw)
 * the table row of its action and (c, s) = cos, sin of its heading in radians (binary32, as torch evaluates them):
 *   angvelocity <- keep * angvelocity + dw;  velocity <- keep * velocity + (c dx - s dy, s dx + c dy)
 * written back to the agents' tensors as the modules do, then the step proceeds.  keep = 1 - decay; keep = 0 assigns. */
typedef struct MsMovement {
    const long long* actions;   /* (N, A) row of `table` per agent (clamped to the table) */
    const float*     table;     /* (n_actions, 3) [dx, dy, dw]: agent-frame velocity and angular velocity deltas */
    int              n_actions;
    float            keep;
} MsMovement;

/* Optional bookkeeping around the physics step: what the reference's environments do with a few dozen tensor ops
 * right before and after it - lifespans (modules.py:328-381), respawns (modules.py:298-326) and the IMU observation
 * (modules.py:240-270).  Every part is optional (NULL pointer = skipped).  Per agent, in this order:
 *   tick     lifespans += 1;  respawn_mask |= lifespans >= max_lifespans;  where the mask is set, lifespans = 0 and
 *            max_lifespans = fresh_max (drawn by the caller, as `actions` are)
 *   respawn  where respawn_mask is set: pose = spawn table row `respawn_choice` (clamped to the table) of this agent,
 *            velocities = 0.  respawn_after = 0: before the movement prologue and the step, so the new pose is what
 *            moves and collides (Deathmatch's order, demo/envs/deathmatch.py:96-99); respawn_after = 1: after the
 *            step's integration (Explorer's order, demo/envs/explorer.py:83-90)
 *   imu      of the state the call leaves behind: [angvelocity/imu_ang_scale, (c vx + s vy)/imu_speed_scale,
 *            (-s vx + c vy)/imu_speed_scale] with (c, s) = cos, sin of the heading in radians, binary32 as torch does - the
 *            quotients included: a tensor over a scalar is `a * (1.f/b)` in ATen */
typedef struct MsStepExtras {
    unsigned char*   respawn_mask;      /* (N, A) bytes, non-zero = respawn; written back when lifespans tick  */
    const long long* respawn_choice;    /* (N, A) */
    const float*     spawn_positions;   /* (N, A, n_spawns, 2) */
    const float*     spawn_angles;      /* (N, A, n_spawns)    */
    int              n_spawns;
    int              respawn_after;
    int*             lifespans;         /* (N, A) */
    int*             max_lifespans;     /* (N, A) */
    const int*       fresh_max;         /* (N, A) */
    float*           imu;               /* (N, A, 3) */
    float            imu_ang_scale, imu_speed_scale;
} MsStepExtras;

/* Replaces `physics(scenery, agents) -> Physics` (wrappers.cpp:69, kernels.cu:179-230):
 * collision-limited integration of the agents, in place; `progress` is the (N, A) output that
 * the reference returns as Physics.progress. */
int ms_move_physics(const MsScenery* scenery, const MsAgents* agents, const MsMovement* movement /* NULL: none */,
                    float* progress, const MsConfig* config, void* hip_stream);
int ms_physics(const MsScenery* scenery, const MsAgents* agents, float* progress,
               const MsConfig* config, void* hip_stream);
/* The same step with the environment's bookkeeping around it in the same launch (movement and extras may be NULL). */
int ms_step_physics(const MsScenery* scenery, const MsAgents* agents, const MsMovement* movement, const MsStepExtras* extras,
                    float* progress, const MsConfig* config, void* hip_stream);

/* Replaces `render(scenery, agents) -> Render` (wrappers.cpp:82, kernels.cu:297-475):
 * draw (rewrites the agent rows of lines_vals) -> raycast -> shade, one fused launch. */
int ms_render(const MsScenery* scenery, const MsAgents* agents, const MsRender* out,
              const MsConfig* config, void* hip_stream);

/* One step of the hot path - ms_physics then ms_render (the reference's every env.step(): wrappers.cpp:69 + :82) - in one
 * call, and where the shapes allow it in ONE LAUNCH: with one agent per env and at most 64 rays (BASELINE config 2: Explorer's
 * shape) the agent is a single wavefront, which runs its env's physics step (kernels.cu:179-230) and renders from the pose it
 * ends on (kernels.cu:297-475); nothing crosses waves, so there is nothing to order between two launches.  Every other shape
 * (several agents per env, more than 64 rays, a wall grid that serves one half of the step only) is ms_physics followed by
 * ms_render, as if the caller had made the two calls.  Same results as the two calls, bit for bit, either way. */
int ms_step_render(const MsScenery* scenery, const MsAgents* agents, float* progress, const MsRender* out,
                   const MsConfig* config, void* hip_stream);
/* ... with the movement prologue and the env's bookkeeping of ms_step_physics around the step (either may be NULL): a whole
 * env.step() of a single-agent env of up to 64 rays - the reference's tutorial env, demo/envs/minimal.py: SimpleMovement, physics,
 * render - is then one launch. */
int ms_move_step_render(const MsScenery* scenery, const MsAgents* agents, const MsMovement* movement, const MsStepExtras* extras,
                        float* progress, const MsRender* out, const MsConfig* config, void* hip_stream);

/* What the reference's Deathmatch env does between one frame and the next - `_reset` + `_shoot` + the `health` observation,
 * megastep/demo/envs/deathmatch.py:46-88: some twenty tensor ops on (N, A) tensors - as one element-wise launch behind
 * ms_render, whose obs_centre it reads.  Per agent-row i = n A + a, in this order:
 *   revive   where `dead` is set (the mask this step's physics launch respawned by): health = 1, damage = 0      (:46-52)
 *   shoot    hits = distinct agents in centre[i][0..1] (ids outside 0 .. A-1 are nobody); wounds = agents b of the env with
 *            a in centre[b][0..1]; outside = position below -clearance or above upper[n] (= extent + clearance) in x or y;
 *            damage += hit_damage * hits;  health += -hit_damage * (wounds + outside) - tick_damage               (:54-72)
 *   report   reset_out = the incoming `dead`; reward = hits; health_obs = health; dead = health <= 0 - the next step's
 *            respawn mask; matchings[i][b] = b in centre[i][0..1]   (each optional: NULL = skipped, except `dead`) */
typedef struct MsDeathmatch {
    const int*      centre;        /* (N, A, 2): MsRender.obs_centre of this frame                              */
    const float*    positions;     /* (N, A, 2): MsAgents.positions                                             */
    const float*    upper;         /* (N, 2): the floorplan's extent (masks.shape * res) + clearance            */
    float           clearance, hit_damage, tick_damage;       /* the reference's: 1, .05, .001                 */
    float*          health;        /* (N, A) in / out                                                           */
    float*          damage;        /* (N, A) in / out                                                           */
    unsigned char*  dead;          /* (N, A) in: revived at this step's start; out: dead now                    */
    unsigned char*  reset_out;     /* (N, A) out, optional                                                      */
    float*          reward;        /* (N, A) out, optional                                                      */
    float*          health_obs;    /* (N, A) out, optional                                                      */
    unsigned char*  matchings;     /* (N, A, A) out, optional                                                   */
} MsDeathmatch;
int ms_deathmatch_shoot(int n_envs, int n_agents, const MsDeathmatch* dm, void* hip_stream);

/* What the reference's Explorer env does between one frame and the next - `_reward`'s arithmetic, `_reset`'s counters and the
 * episode rule of `step`, megastep/demo/envs/explorer.py:45-90 - as one launch of N threads behind ms_render, whose first-sight
 * tally (MsRender.seen_count; the stamps and epochs are MsRender.seen_stamp / seen_epoch) it reads.  Per env n, in this order:
 *   reward   reset_out = over (this step began with a respawn); reward = over ? 0 : (tally - before) / pixels;
 *            potential = tally, length_out = lengths - as this step leaves them (for display; optional)
 *   next     lengths += 1;  over = lengths >= tally + slack - the NEXT step's respawn mask (MsStepExtras.respawn_mask,
 *            respawn_after = 1) - and where it is set: epoch += 1 (the env forgets every texel at once), tally = lengths = 0;
 *            before = tally */
typedef struct MsExplorer {
    int*            tally;         /* (N) in / out: MsRender.seen_count                                          */
    int*            before;        /* (N) in / out: tally at the last reward                                      */
    int*            lengths;       /* (N) in / out: episode lengths                                               */
    int*            epoch;         /* (N) in / out: MsRender.seen_epoch                                           */
    unsigned char*  over;          /* (N) in: respawned this step; out: to be respawned by the next               */
    int             slack;         /* steps an episode lasts on top of one per texel seen (the reference's 200)   */
    int             pixels;        /* observation pixels per agent: res / subsample                               */
    unsigned char*  reset_out;     /* (N) out, optional                                                           */
    float*          reward;        /* (N) out                                                                     */
    float*          potential;     /* (N) out, optional                                                           */
    int*            length_out;    /* (N) out, optional                                                           */
} MsExplorer;
int ms_explorer_books(int n_envs, const MsExplorer* books, void* hip_stream);

/* Builds the wall grid (MsScenery.wg_*): per level of cells two launches with a prefix sum by the caller in between.
 *   ms_wallgrid_scan  for every cell of every env listed in `reps` (the representatives, MsScenery.env_geom; n_reps of
 *                     them) works out which static walls belong on the cell's lists: one bit per wall into `bits` - the
 *                     row of cell c (grid-local) of env n and kind k (0 vis, 1 near within wg_reach_lo, 2 near beyond that)
 *                     starts at word bits_starts[n] + (3 c + k) * ceil(walls(n)/32) - and the three counts into
 *                     counts[3*(wg_starts[n] + c) + k].  Reads wg_starts, wg_geom, wg_cell, wg_reach_lo, wg_reach, wg_near
 *                     of the scenery; `bits` and `counts` must start zeroed.
 *                     `parent`: a coarser grid over the same envs, scanned and filled before (same origin, cells a whole
 *                     multiple of wg_cell in size, near lists as indices): only what is on a parent cell's lists is looked
 *                     at for the cells inside it - exact (a wall hidden from, or out of reach of, the larger cell is so for
 *                     every cell within) and an order of magnitude less work on large floorplans.  NULL: every wall.
 *                     max_groups: with a parent the most parent cells any listed env has, without ceil(most cells / 4).
 *   ms_wallgrid_fill  writes the lists: the set bits of each row, in order, from the cell's wg_cells offsets (which the
 *                     caller has filled in from the counts) - vis lists as entries of `vis_entries` (MsScenery.wg_pool's
 *                     format) from the env's wg_pool_base on (which the scenery must carry then), near lists as rows into
 *                     `near_rows`; or, for a parent level (both NULL; wg_pool_base is not looked at), both lists as
 *                     16-bit indices into `pool`.
 * An env with more than 65535 static walls must have a grid of 0 cells. */
#define MS_WALLGRID_MAX_FOV 165.f
typedef struct MsWallGridParent {
    const unsigned* cells; const int* starts; const float* geom; float cell; const unsigned short* pool;
} MsWallGridParent;
int ms_wallgrid_scan(const MsScenery* scenery, const MsWallGridParent* parent, const int* reps, int n_reps, int max_groups,
                     const long long* bits_starts, unsigned* bits, unsigned* counts, void* hip_stream);
int ms_wallgrid_fill(const MsScenery* scenery, const int* reps, int n_reps, int max_cells,
                     const long long* bits_starts, const unsigned* bits, unsigned short* pool, unsigned* vis_entries,
                     float* near_rows, void* hip_stream);
/* The host instantiations of the kernels' culls (ms_host_*) and the debug switches (ms_debug_*), which exist for the CPU
 * tests and A/B runs only, are declared in megastep_hip_test.h: not part of the interface a maintainer binds. */

#ifdef __cplusplus
}
#endif
#endif /* MEGASTEP_HIP_H */
