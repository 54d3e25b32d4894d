"""Random worlds against the oracle: shapes, views and speeds drawn per seed - agents per env 1..70, rays 1..600, fov
20..175 degrees, small and large floorplans, toys - several physics + render steps each, with the movement / respawn
extras and the pooled observations switched on at random; every second step goes through ms_step_render (the one-launch step
where the shape allows it). Prints one line per seed and a summary; exits 1 on a mismatch.
usage: python tools/fuzz_parity.py [first_seed] [n_seeds]"""
import sys, time, traceback
sys.path.insert(0, '.')
import numpy as np, torch
from tests import util
from megastep_amd import core, cubicasa, cuda, scene, toys


def one(seed, oblique=None):
    """`oblique` (None: every third seed): floorplans turned by seeded angles with diagonal partitions (cubicasa.sample(oblique=True):
    the reference's walls are exteriors of arbitrary polygons, geometry.py:43-57), and, where walls are mutated, some of them
    swung about their middles by random angles as well."""
    rng = np.random.RandomState(1000 + seed)
    forced = oblique is True
    oblique = (seed % 3 == 2) if oblique is None else oblique
    n_agents = int(rng.choice([1, 1, 2, 3, 4, 4, 5, 8, 17, 65, 70], p=[.15, .1, .15, .1, .15, .1, .1, .06, .05, .02, .02]))
    res = int(rng.choice([1, 3, 8, 64, 64, 100, 128, 256, 512, 600]))
    if n_agents > 8:
        res = min(res, 128)
    fov = float(rng.choice([20, 70, 90, 130, 130, 170, 175]))
    kind = rng.choice(['plans', 'plans', 'plans', 'large', 'box', 'column'])
    if forced and kind in ('box', 'column'):
        kind = 'plans'
    n_envs = int(rng.randint(1, 4 if kind == 'large' or n_agents > 8 else 12))
    np.random.seed(seed)
    if kind in ('box', 'column'):
        geometries = n_envs*[getattr(toys, kind)()]
    else:
        geometries = cubicasa.sample(n_envs, n_unique=16, seed=seed + 1, large=kind == 'large', oblique=oblique)
        kind = kind + ('/o' if oblique else '')
    sc = scene.scenery(geometries, n_agents, device='cuda', random=np.random.RandomState(seed))
    c = core.Core(sc, res=res, fov=fov, fps=float(rng.choice([10, 10, 30, 3])))
    util.spawn(c, geometries, seed=seed)
    mutated = ''
    if seed % 2:
        # walls moved onto each other, collapsed to points, stretched: ties for the 1e-4 rule, degenerate segments
        AF = n_agents*sc.model.shape[0]
        lines = sc.lines.vals.reshape(-1, 4)
        starts, widths = sc.lines.starts.cpu().numpy(), sc.lines.widths.cpu().numpy()
        for e in range(n_envs):
            walls = np.arange(starts[e] + AF, starts[e] + widths[e])
            if len(walls) < 2:
                continue
            for _ in range(max(2, len(walls)//8)):
                i, j = rng.choice(walls, 2, replace=False)
                how = rng.randint(6 if oblique else 5)
                if how == 5:                                                         # swung about its middle
                    mid, half = (lines[i, :2] + lines[i, 2:])/2, (lines[i, 2:] - lines[i, :2])/2
                    th = float(rng.uniform(0, np.pi))
                    half = torch.stack([np.cos(th)*half[0] - np.sin(th)*half[1], np.sin(th)*half[0] + np.cos(th)*half[1]])
                    lines[i] = torch.cat([mid - half, mid + half])
                elif how == 0:   lines[i] = lines[j]                                   # coincident
                elif how == 1: lines[i] = lines[j][[2, 3, 0, 1]]                     # coincident, reversed
                elif how == 2: lines[i, 2:] = lines[i, :2]                           # a point
                elif how == 3: lines[i] = lines[j] + float(rng.choice([2e-5, 9e-5, 1.1e-4, 1e-3]))   # inside / outside the band
                else:          lines[i, 2:] = lines[i, :2] + 100*(lines[i, 2:] - lines[i, :2])       # very long
        # (the envs no longer share floorplans: a scenery of its own, without the build's `geom` table)
        sc = cuda.Scenery(n_agents, sc.lights, sc.lines, sc.textures, sc.model)
        cuda.bake(sc)
        c.scenery = sc
        mutated = ' mutated'
    ref = util.OracleWorld(c)
    np.testing.assert_allclose(c.scenery.baked.vals.cpu().numpy(), ref.bake(), rtol=0, atol=1e-5)
    ref.pull_baked(c)
    fields = cuda.FIELDS
    for step in range(3):
        util.random_velocities(c, rng, speed=float(rng.choice([.01, 1., 4., 40.])), spin=float(rng.choice([0., 90., 720.])))
        if step == 2:            # some agents standing still, one crawling
            c.agents.velocity[::2] = 0.
            c.agents.velocity[0, 0] = torch.tensor([3e-6, 0.], device='cuda')
        ref.pull_agents(c)
        if step % 2:
            # (through ms_step_render: ONE launch for a single agent of up to 64 rays, the two launches for every other shape)
            p, r = cuda.step_render(c.scenery, c.agents)
        else:
            p = cuda.physics(c.scenery, c.agents)
            r = cuda.render(c.scenery, c.agents)
        prog_ref, agents_ref = ref.physics()
        util.assert_physics_matches(c, p, prog_ref, agents_ref)
        util.assert_render_matches(c, r, ref.render())
    return f'{kind:8s} envs {n_envs:2d} agents {n_agents:2d} rays {res:3d} fov {fov:5.1f}{mutated}'


def one_fused(seed):
    """What the render kernel can write besides the reference's five planes - pooled RGB-D, the crosshair ids, the
    first-sight books, any subset of the planes - against tensor ops on a full render of the same state."""
    rng = np.random.RandomState(5000 + seed)
    n_agents = int(rng.choice([1, 2, 3, 4, 6]))
    sub = int(rng.choice([1, 2, 4, 8, 16, 32, 64]))
    px = int(rng.randint(2, 13))
    res = sub*px
    n_envs = int(rng.randint(1, 9))
    np.random.seed(seed)
    geometries = cubicasa.sample(n_envs, n_unique=16, seed=seed + 1) if rng.rand() < .7 else n_envs*[toys.column()]
    sc = scene.scenery(geometries, n_agents, device='cuda', random=np.random.RandomState(seed))
    c = core.Core(sc, res=res, fov=float(rng.choice([70, 130, 170])), fps=10)
    util.spawn(c, geometries, seed=seed)
    util.random_velocities(c, rng, speed=4.)
    if rng.rand() < .7:
        cuda.physics(sc, c.agents)                       # (leaves the heading cache for the renders)
    full = cuda.render(sc, c.agents)
    M, AF = sc.model.shape[0], n_agents*sc.model.shape[0]
    max_depth = float(rng.choice([3., 10.]))
    want_rgb = full.screen.reshape(n_envs, n_agents, px, sub, 3).mean(3).permute(0, 1, 3, 2)
    want_d = (1 - ((full.distances - c.agent_radius)/max_depth).clamp(0, 1)).reshape(n_envs, n_agents, px, sub).mean(3)
    r1, r2 = (px//2 - 1)*sub + sub//2, (px//2)*sub + sub//2
    idx = full.indices[..., [r1, r2]]
    want_centre = torch.where((idx >= 0) & (idx < AF), idx//M, torch.full_like(idx, -1))
    hit = full.indices >= 0
    line = (sc.lines.starts[:, None, None] + full.indices.clamp(min=0)).long()
    width = sc.textures.widths[line].float()
    along = torch.min(torch.floor(width*full.locations), width - 1)
    texel = sc.textures.starts[line].long() + torch.where(hit, along, torch.zeros_like(along)).long()
    want_stamp = torch.zeros(sc.textures.vals.shape[0], dtype=torch.int32, device='cuda')
    want_stamp[texel[hit]] = 1
    want_count = torch.stack([texel[e][hit[e]].unique().numel()*torch.ones((), dtype=torch.int32, device='cuda') for e in range(n_envs)])
    if not hit.all():
        # a ray that missed marks the scenery's LAST texel, to the credit of the last env (explorer.py:36,47: `_seen[-1] = True`)
        if want_stamp[-1] == 0:
            want_count[-1] += 1
        want_stamp[-1] = 1
    fields = tuple(f for f in cuda.FIELDS if rng.rand() < .4)
    books = (torch.zeros_like(want_stamp), torch.ones(n_envs, dtype=torch.int32, device='cuda'), torch.zeros(n_envs, dtype=torch.int32, device='cuda'))
    fused = cuda.render(sc, c.agents, fields=fields, pooled=dict(subsample=sub, max_depth=max_depth, centre=True), seen=books)
    for f in cuda.FIELDS:
        if f in fields:
            assert torch.equal(torch.nan_to_num(getattr(fused, f).float(), nan=-7.), torch.nan_to_num(getattr(full, f).float(), nan=-7.)), f
        else:
            assert getattr(fused, f) is None, f
    torch.testing.assert_close(fused.obs_rgb, want_rgb, rtol=0, atol=1e-6)
    torch.testing.assert_close(fused.obs_depth, want_d, rtol=0, atol=1e-6)
    assert torch.equal(fused.obs_centre, want_centre.int())
    assert torch.equal(books[0], want_stamp) and torch.equal(books[2], want_count)
    again = cuda.render(sc, c.agents, fields=fields, pooled=dict(subsample=sub, max_depth=max_depth, centre=True), seen=books)
    assert torch.equal(books[2], want_count), 'a second look adds nothing'
    return f'fused  envs {n_envs:2d} agents {n_agents:2d} rays {res:3d} = {px} px x {sub} fields {",".join(fields) or "-"}'


if __name__ == '__main__':
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    bad, t0 = [], time.time()
    for seed in range(first, first + n):
        try:
            print(seed, one(seed), 'ok', flush=True)
            print(seed, one_fused(seed), 'ok', flush=True)
        except Exception as e:
            bad.append(seed)
            print(seed, 'MISMATCH', type(e).__name__, str(e)[:400].replace('\n', ' | '), flush=True)
            traceback.print_exc(limit=3)
    bad = sorted(set(bad))
    print(f'{n - len(bad)}/{n} seeds agree with the oracle in {time.time() - t0:.0f} s; mismatches: {bad}')
    sys.exit(1 if bad else 0)
