// kernels/envlogic.h -- deathmatch_kernel: what the reference's Deathmatch env does with a few dozen tensor ops between
// one frame and the next (reference: megastep/demo/envs/deathmatch.py:46-72 `_reset` + `_shoot`, :74-88 the `health`
// observation), as one element-wise launch.  Part of megastep_hip.hip's one translation unit.
//
// A Deathmatch step is: respawn the dead (in the physics launch: MsStepExtras), move, step (physics), look (render, which
// already leaves `obs_centre` - the agent in each of every agent's two central observation pixels, or -1), then settle the
// frame's fire: who has whom in the crosshair, hits and wounds, agents strayed outside their floorplan, health, damage, the
// reward, and who will be dead at the start of the next step.  On (4096, 4) tensors each of those ~20 torch kernels lasts
// 4-6 us - 40 % of a 270 us step (profiles/r05_env_deathmatch_kernel_stats.csv); this is ONE launch of N x A threads.
//
// Arithmetic: the reference's, statement by statement, in binary32 without contraction -
//     damage += .05 hits;      health += -.05 (wounds + outside) - .001          (deathmatch.py:64,70)
// with hits = the number of distinct agents in this agent's two centre pixels (matchings.sum(2)), wounds = the number of
// agents that have this one in either of theirs (matchings.sum(1)), outside = any coordinate of the position below -clearance
// or above the floorplan's extent + clearance (:66-67).  Before that, the agents the step began by reviving (`dead`, as the
// physics launch's respawn mask saw it) get health 1 and damage 0 (:46-52) - the fire exchange then works on those values, as
// it does in the reference, where `_reset` runs at the top of `step`.

// lane = agent-row i = n A + a
__global__ __launch_bounds__(WG) void deathmatch_kernel(const MsDeathmatch dm, const int n_envs, const int n_agents) {
    const long long i = (long long)blockIdx.x*WG + threadIdx.x;
    if (i >= (long long)n_envs*n_agents) return;
    const int n = (int)(i / n_agents), a = (int)(i - (long long)n*n_agents);
    const int2* __restrict__ centre = reinterpret_cast<const int2*>(dm.centre) + (long long)n*n_agents;
    const int2 mine = centre[a];
    // hits: distinct agents in my two centre pixels (an id outside 0 .. A-1 - the -1 of "no agent" - is nobody)
    const bool ok0 = (mine.x >= 0) & (mine.x < n_agents), ok1 = (mine.y >= 0) & (mine.y < n_agents);
    const float hits = (float)((ok0 ? 1 : 0) + ((ok1 & !(ok0 & (mine.x == mine.y))) ? 1 : 0));
    // wounds: agents that have me in either of theirs
    int wounds_i = 0;
    for (int b = 0; b < n_agents; b++) {
        const int2 c = centre[b];
        wounds_i += ((c.x == a) | (c.y == a)) ? 1 : 0;
    }
    const float2 p = reinterpret_cast<const float2*>(dm.positions)[i];
    const float2 up = reinterpret_cast<const float2*>(dm.upper)[n];
    const bool outside = (p.x < -dm.clearance) | (p.y < -dm.clearance) | (p.x > up.x) | (p.y > up.y);
    const bool revived = dm.dead[i] != 0;
    float health = revived ? 1.f : dm.health[i];
    float damage = revived ? 0.f : dm.damage[i];
    damage = damage + dm.hit_damage*hits;
    health = health + (-dm.hit_damage*((float)wounds_i + (outside ? 1.f : 0.f)) - dm.tick_damage);
    dm.health[i] = health;
    dm.damage[i] = damage;
    if (dm.reset_out) dm.reset_out[i] = revived ? 1 : 0;
    if (dm.reward) dm.reward[i] = hits;
    if (dm.health_obs) dm.health_obs[i] = health;
    dm.dead[i] = (health <= 0.f) ? 1 : 0;                 // (NaN - an env that was never reset - is not dead: `nan <= 0` is false in torch too)
    if (dm.matchings) {
        unsigned char* row = dm.matchings + i*n_agents;
        for (int b = 0; b < n_agents; b++) row[b] = ((mine.x == b) | (mine.y == b)) ? 1 : 0;
    }
}

// explorer_kernel: the Explorer env's bookkeeping between one frame and the next (reference: megastep/demo/envs/explorer.py:45-58
// `_reward`, :68-72 `_reset`'s counters, :83-90 the episode rule in `step`) - a dozen tensor ops on (N,) tensors, 35 us of a 79 us
// step (profiles/r05_env_explorer_kernel_stats.csv) - as one launch of N threads behind ms_render, whose first-sight tally
// (MsRender.seen_count) it reads.  Per env: this frame's reward - the texels seen for the first time, per observation pixel; none
// for a frame that began with a respawn - and then, ahead of time, the NEXT step's episode rule: the length ticks, an env whose
// length has reached its tally + slack is marked over (the mask the next physics launch respawns by, MsStepExtras) and forgets
// what it has seen (its epoch moves on, its counters start over) - exactly what the reference does at the top of that step;
// nothing in between looks at these counters.
__global__ __launch_bounds__(WG) void explorer_kernel(const MsExplorer ex, const int n_envs) {
    const int n = blockIdx.x*WG + threadIdx.x;
    if (n >= n_envs) return;
    const bool fresh = ex.over[n] != 0;                  // this step began with a respawn
    int tally = ex.tally[n], length = ex.lengths[n];
    const int gained = tally - ex.before[n];
    if (ex.reset_out) ex.reset_out[n] = fresh ? 1 : 0;
    ex.reward[n] = fresh ? 0.f : (float)gained/(float)ex.pixels;     // explorer.py:52-56 (an int tensor over an int: true division in binary32)
    if (ex.potential) ex.potential[n] = (float)tally;    // (as this step leaves them, for Explorer.state())
    if (ex.length_out) ex.length_out[n] = length;
    // ---- the next step's top (explorer.py:86-89)
    length += 1;
    const bool over = length >= tally + ex.slack;
    if (over) { ex.epoch[n] += 1; tally = 0; length = 0; }
    ex.tally[n] = tally;
    ex.before[n] = tally;
    ex.lengths[n] = length;
    ex.over[n] = over ? 1 : 0;
}
