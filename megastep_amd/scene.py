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

def _ragged(rows, widths, device):
    vals = arrdict.torchify(np.concatenate(rows)).to(device).contiguous()
    widths = arrdict.torchify(np.asarray(widths).reshape(-1)).to(device).contiguous()
    return ragged.Ragged(vals, widths)


@torch.no_grad()
def scenery(geometries, n_agents=1, device='cuda', random=np.random, bake=True):
    """One env per geometry (dicts with ``walls`` (W, 2, 2) and ``lights`` (I, 2)), ``n_agents`` agents in each, on
    ``device``, lighting baked. ``bake=False`` skips the GPU bake, for host-only plumbing."""
    model = agent_model()
    agentlines, agentcolors = np.tile(model, (n_agents, 1, 1)), np.tile(agent_colors(), (n_agents, 1))

    per_env = []
    for g in geometries:
        lights = random_lights(g['lights'])              # from the GLOBAL np.random, as the reference does (scene.py:82)
        texels, counts = init_textures(agentlines, agentcolors, g['walls'], random)
        per_env.append((lights, np.concatenate([agentlines, g['walls']]), texels, counts))
    lights, lines, texels, counts = zip(*per_env)

    result = cuda.Scenery(
        n_agents=n_agents,
        lights=_ragged(lights, [len(x) for x in lights], device),
        lines=_ragged(lines, [len(x) for x in lines], device),
        textures=_ragged(texels, np.concatenate(counts), device),
        model=arrdict.torchify(model).to(device))
    if bake:
        cuda.bake(result)
    return result
