"""Geometries -> :class:`~megastep_amd.cuda.Scenery`: lines, lights and textures of every env packed into the ragged
tensors the kernels read, with the static lighting baked (behaviour of megastep/scene.py:9-100).

What has to agree with the reference value for value - the agent model, the wall palette, the texel counts and, above
all, the *order in which random numbers are drawn* (per geometry: light intensities from the global ``np.random``,
then one ``choice`` and one ``normal`` for the wall pattern from ``random``) - is pinned by
``tests/golden/reference_host.npz``; a seeded build reproduces the reference's textures and lights exactly."""
import numpy as np
import torch

from . import arrdict, core, cuda, ragged

#: the reference's ten wall colours (scene.py:9-20), sRGB 0-255; wall k of an env wears colour k mod 10
WALL_PALETTE = np.array([
    (0xc1, 0x85, 0xae), (0x73, 0xa1, 0x71), (0x56, 0x66, 0xa4), (0x9f, 0x7c, 0x4a), (0x80, 0x9c, 0xd5),
    (0x56, 0x6e, 0x40), (0x8e, 0x53, 0x7b), (0x4f, 0x9f, 0xa4), (0xb5, 0x6d, 0x66), (0x5a, 0x72, 0x8c)])/255

#: the agent's outline, counter-clockwise from its back-right corner, in units of half an agent width
_OUTLINE = np.array([[-.5, -1.], [+.5, -1.], [+1., -.5], [+1., +.5], [+.5, +1.], [-.5, +1.], [-1., +.5], [-1., -.5]])
#: colour of the outline's segments: dark flanks, green nose and tail, red cheeks (scene.py:35-38)
_DARK, _GREEN, _RED = (.25, .25, .25), (0., .5, 0.), (1., 0., 0.)
_LIVERY = np.array([_DARK, _GREEN, _DARK, _RED, _DARK, _RED, _DARK, _GREEN])


# ---- the agent ----------------------------------------------------------------------------------------------------

def agent_model():
    """(8, 2, 2): the segments of an agent's outline in its own frame, nose along +x."""
    return core.AGENT_WIDTH/2*np.stack([_OUTLINE, np.roll(_OUTLINE, -1, axis=0)], axis=1)


def agent_colors():
    """(8, 3): linear-light-to-be RGB of each outline segment."""
    return _LIVERY.copy()


# ---- texels ---------------------------------------------------------------------------------------------------------

def lengths(lines):
    """Euclidean length of (..., 2, 2) segments."""
    delta = lines[..., 0, :] - lines[..., 1, :]
    return (delta**2).sum(-1)**.5


def resolutions(lines):
    """Texels per segment: one per ``core.TEXTURE_RES`` metres, rounded up."""
    return np.ceil(lengths(lines)/core.TEXTURE_RES).astype(int)


def wall_pattern(n, l=.5, random=np.random):
    """``n`` brightness values in [.5, 1): a level that holds for ~``l`` metres of texels, then jumps - stripes that
    make depth readable. Draws ``choice`` then ``normal``, ``n`` each (scene.py:43-48)."""
    jump_probability = core.TEXTURE_RES/l
    jumps = random.choice(np.array([0., 1.]), p=np.array([1 - jump_probability, jump_probability]), size=n)
    jumps = jumps*random.normal(size=n)
    return .5 + .5*(jumps.cumsum() % 1)


def init_textures(agentlines, agentcolors, walls, random=np.random):
    """Texels of one env: ((sum T, 3) linear RGB, (L,) texels per line), agent lines first. Agents are drawn flat,
    walls in their palette colour under a :func:`wall_pattern`."""
    counts = resolutions(np.concatenate([agentlines, walls]))
    per_line = np.concatenate([agentcolors, WALL_PALETTE[np.arange(len(walls)) % len(WALL_PALETTE)]])
    texels = core.gamma_decode(np.repeat(per_line, counts, axis=0))
    brightness = wall_pattern(len(texels), random=random)
    brightness[:counts[:len(agentlines)].sum()] = 1.
    return texels*brightness[:, None], counts


def random_lights(lights, random=np.random):
    """(I, 2) light positions -> (I, 3) with an intensity drawn from U(.5, 2) appended."""
    intensity = random.uniform(.5, 2., (len(lights), 1))
    return np.concatenate([lights, intensity], axis=-1)


# ---- the scenery ----------------------------------------------------------------------------------------------------
# The reference assembles every env in a Python loop - repeat, gamma-decode and concatenate per geometry
# (scene.py:75-100) - which is fine for hundreds of envs and hopeless for the 32 768 per GPU of the large-map
# benchmark point. Here everything that depends on the floorplan alone (lines, texel counts, texel colours) is worked
# out once per DISTINCT geometry, vectorised over all of them, and expanded to envs by gathers on the device; only what
# the reference draws per env from its random streams - light intensities and wall patterns - is per env. Envs built
# from one geometry object are also marked as sharing it (`Scenery.geom`), so `bake` lights each floorplan once.

#: texels handled per device chunk when texturing (bounds the float64 temporaries)
_TEXEL_CHUNK = 1 << 26


def _distinct(geometries):
    """Geometries by identity: (the distinct ones in order of first appearance, env -> distinct index)."""
    seen, distinct, which = {}, [], np.empty(len(geometries), np.int64)
    for n, g in enumerate(geometries):
        u = seen.get(id(g))
        if u is None:
            u = seen[id(g)] = len(distinct)
            distinct.append(g)
        which[n] = u
    return distinct, which


def _ragged_arange(widths):
    """[0..w0), [0..w1), ... back to back."""
    widths = np.asarray(widths, np.int64)
    ends = widths.cumsum()
    return np.arange(ends[-1] if len(ends) else 0) - np.repeat(ends - widths, widths)


def _floorplan_tables(distinct, agentlines, agentcolors):
    """Per distinct geometry, back to back: lines (agent lines first), texels per line, decoded colour per line,
    and the widths (lines, lights) that delimit them."""
    n_walls = np.array([len(g['walls']) for g in distinct], np.int64)
    af = len(agentlines)
    walls = np.concatenate([np.asarray(g['walls'], float).reshape(-1, 2, 2) for g in distinct]) if len(distinct) else np.zeros((0, 2, 2))
    n_lines = af + n_walls
    is_agent = _ragged_arange(n_lines) < af
    lines = np.empty((n_lines.sum(), 2, 2))
    lines[is_agent] = np.tile(agentlines, (len(distinct), 1, 1))
    lines[~is_agent] = walls
    colours = np.empty((n_lines.sum(), 3))
    colours[is_agent] = np.tile(agentcolors, (len(distinct), 1))
    colours[~is_agent] = WALL_PALETTE[_ragged_arange(n_walls) % len(WALL_PALETTE)]
    lights = [np.asarray(g['lights'], float).reshape(-1, 2) for g in distinct]
    n_lights = np.array([len(x) for x in lights], np.int64)
    lights = np.concatenate(lights) if len(lights) else np.zeros((0, 2))
    return lines, resolutions(lines), core.gamma_decode(colours), n_lines, lights, n_lights


def _expand(widths_u, which, device):
    """For envs built from distinct blocks `which`: (widths per env, for every row of the expansion the row of the
    distinct table it copies), as device tensors."""
    widths_u = torch.as_tensor(widths_u, device=device)
    which = torch.as_tensor(which, device=device)
    starts_u = widths_u.cumsum(0) - widths_u
    widths = widths_u[which]
    starts = widths.cumsum(0) - widths
    total = int(widths.sum())
    src = torch.arange(total, device=device) + torch.repeat_interleave(starts_u[which] - starts, widths, output_size=total)
    return widths, src


def _device_pattern(n_texels, starts, lengths, device, l=.5):
    """`wall_pattern` for many envs at once, drawn on the device: texel t of the chunk belongs to the env whose
    [start, start + length) holds it. Same distribution as the reference's, not its random stream."""
    jump_probability = core.TEXTURE_RES/l
    jumps = (torch.rand(n_texels, device=device, dtype=torch.float64) < jump_probability)*torch.randn(n_texels, device=device, dtype=torch.float64)
    total = jumps.cumsum(0)
    before = torch.cat([total.new_zeros(1), total])[starts]            # the running total where each env begins
    level = total - torch.repeat_interleave(before, lengths, output_size=n_texels)
    return .5 + .5*(level % 1)


@torch.no_grad()
def scenery(geometries, n_agents=1, device='cuda', random=np.random, bake=True, fast=False, envs=None):
    """One env per geometry (dicts with ``walls`` (W, 2, 2) and ``lights`` (I, 2)), ``n_agents`` agents in each, on
    ``device``, lighting baked. ``bake=False`` skips the GPU bake, for host-only plumbing.

    ``envs=(start, stop)`` builds only that contiguous slice of the world ``geometries`` describes - what one GPU of
    several owns (reference: common.h:136-144 slices, it does not replicate) - and nothing of the rest reaches the
    device; the random streams are still advanced past the envs before ``start``, so the slice holds exactly the rows
    the whole build would (``tests/test_bench_gloo.py``). With ``fast=True`` there is no stream to keep in step.

    Light intensities and wall patterns come from the reference's random streams in the reference's order (per env:
    intensities from the global ``np.random``, then the pattern's ``choice`` and ``normal`` from ``random``), so a
    seeded build reproduces the reference's scenery value for value. ``fast=True`` draws them on the device from
    torch's generator instead - same distributions, a different stream - which is what makes 10^4..10^5 envs a
    matter of seconds."""
    geometries = list(geometries)
    model = agent_model()
    agentlines, agentcolors = np.tile(model, (n_agents, 1, 1)), np.tile(agent_colors(), (n_agents, 1))
    distinct, which = _distinct(geometries)
    lines_u, counts_u, colours_u, n_lines_u, lights_u, n_lights_u = _floorplan_tables(distinct, agentlines, agentcolors)
    start, stop = (0, len(geometries)) if envs is None else (int(envs[0]), int(envs[1]))
    if not (0 <= start <= stop <= len(geometries)):
        raise ValueError(f'envs={envs} is not a slice of {len(geometries)} geometries')
    which_all, which = which, which[start:stop]

    # lines and texel counts of every env: gathers from the distinct tables
    line_widths, line_src = _expand(n_lines_u, which, device)
    lines = ragged.Ragged(arrdict.torchify(lines_u).to(device)[line_src].contiguous(), line_widths.to(torch.int32))
    texel_widths = torch.as_tensor(counts_u, device=device)[line_src].to(torch.int32)

    # texels per distinct geometry and per env
    line_ends_u = n_lines_u.cumsum()
    texel_ends_u = np.concatenate([[0], counts_u.cumsum()])
    texels_u = texel_ends_u[line_ends_u] - texel_ends_u[line_ends_u - n_lines_u]
    n_texels, n_lights = texels_u[which], n_lights_u[which]
    agent_texels = int(resolutions(agentlines).sum())
    n_envs = stop - start

    # the per-env random part
    if fast:
        intensity = torch.empty(int(n_lights.sum()), device=device, dtype=torch.float64).uniform_(.5, 2.)
        brightness = None
    else:
        intensity, brightness = np.empty(n_lights.sum()), np.empty(n_texels.sum())
        for u in which_all[:start]:                                     # the envs before the slice: their draws, thrown away
            np.random.uniform(.5, 2., (n_lights_u[u], 1))
            wall_pattern(texels_u[u], random=random)
        i0 = t0 = 0
        for ni, nt in zip(n_lights, n_texels):
            intensity[i0:i0 + ni] = np.random.uniform(.5, 2., (ni, 1))[:, 0]   # GLOBAL np.random, as scene.py:82 does
            pattern = wall_pattern(nt, random=random)
            pattern[:agent_texels] = 1.
            brightness[t0:t0 + nt] = pattern
            i0, t0 = i0 + ni, t0 + nt
        intensity = torch.as_tensor(intensity, device=device)

    light_widths, light_src = _expand(n_lights_u, which, device)
    positions = torch.as_tensor(lights_u, device=device)[light_src]
    lights = ragged.Ragged(torch.cat([positions, intensity[:, None]], 1).float().contiguous(), light_widths.to(torch.int32))

    # texel colours: (decoded colour of the texel's line) x (brightness of the texel), in float64 like the reference,
    # rounded to float32 once; a chunk of envs at a time to bound the temporaries
    textures = ragged.Ragged(torch.empty((int(n_texels.sum()), 3), device=device), texel_widths)
    colours = torch.as_tensor(colours_u, device=device)
    texel_ends = np.concatenate([[0], n_texels.cumsum()])
    e0 = 0
    while e0 < n_envs:
        e1 = max(int(np.searchsorted(texel_ends, texel_ends[e0] + _TEXEL_CHUNK, 'right')) - 1, e0 + 1)
        t0, t1 = int(texel_ends[e0]), int(texel_ends[e1])
        if brightness is None:
            starts = torch.as_tensor(texel_ends[e0:e1] - t0, device=device)
            lengths = torch.as_tensor(n_texels[e0:e1], device=device)
            shade = _device_pattern(t1 - t0, starts, lengths, device)
            is_agent = (torch.arange(t1 - t0, device=device) - torch.repeat_interleave(starts, lengths, output_size=t1 - t0)) < agent_texels
            shade = torch.where(is_agent, torch.ones_like(shade), shade)
        else:
            shade = torch.as_tensor(brightness[t0:t1], device=device)
        source_line = line_src[textures.inverse[t0:t1].long()]
        textures.vals[t0:t1] = (colours[source_line]*shade[:, None]).float()
        e0 = e1

    geom = None
    if len(np.unique(which)) < n_envs:
        first_env = np.full(len(distinct), n_envs, np.int64)
        np.minimum.at(first_env, which, np.arange(n_envs))
        geom = torch.as_tensor(first_env[which], device=device).to(torch.int32)
    result = cuda.Scenery(n_agents=n_agents, lights=lights, lines=lines, textures=textures,
                          model=arrdict.torchify(model).to(device), geom=geom)
    if bake:
        cuda.bake(result)
    return result
