"""Geometries -> :class:`~megastep_amd.cuda.Scenery` (reference: megastep/scene.py:9-100).

The numpy RNG is consumed in exactly the reference's order (per geometry: light intensities from the *global*
``np.random``, then ``choice`` and ``normal`` for the wall pattern from ``random``), so a seeded build reproduces the
reference's textures and lights value for value."""
import numpy as np
import torch
from . import core, ragged, arrdict, cuda

# Ten bland colours, as in the reference (scene.py:9-20)
COLORS = ["#c185ae", "#73a171", "#5666a4", "#9f7c4a", "#809cd5", "#566e40", "#8e537b", "#4f9fa4", "#b56d66", "#5a728c"]


def _hex_rgb(h):
    return [int(h[i:i + 2], 16)/255 for i in (1, 3, 5)]


def lengths(lines):
    return ((lines[..., 0, :] - lines[..., 1, :])**2).sum(-1)**.5


def agent_model():
    """The 8-segment outline of an agent in its own frame, (8, 2, 2), front along +x (reference: scene.py:25-33)."""
    corners = np.array([[-.5, -1.], [+.5, -1.], [+1., -.5], [+1., +.5], [+.5, +1.], [-.5, +1.], [-1., +.5], [-1., -.5]])
    walls = np.stack([corners, np.roll(corners, -1, 0)], 1)
    return core.AGENT_WIDTH/2*walls


def agent_colors():
    """Per-segment RGB of the agent model: dark sides, green front/back, red left/right (reference: scene.py:35-38)."""
    k, g, r = [.25, .25, .25], [0., .5, 0.], [1., 0., 0.]
    return np.array([k, g, k, r, k, r, k, g])


def resolutions(lines):
    """Texels per line at TEXTURE_RES metres per texel (reference: scene.py:40-41)."""
    return np.ceil(lengths(lines)/core.TEXTURE_RES).astype(int)


def wall_pattern(n, l=.5, random=np.random):
    """A brightness pattern that jumps every ~``l`` metres, to make depth perception easy (reference: scene.py:43-48)."""
    p = core.TEXTURE_RES/l
    jumps = random.choice(np.array([0., 1.]), p=np.array([1 - p, p]), size=n)
    jumps = jumps*random.normal(size=n)
    return .5 + .5*(jumps.cumsum() % 1)


def init_textures(agentlines, agentcolors, walls, random=np.random):
    """(sum T, 3) linear-RGB texels and the (L,) texel count of each line (reference: scene.py:50-68)."""
    palette = np.array([_hex_rgb(c) for c in COLORS])
    colors = np.concatenate([agentcolors, palette[np.arange(len(walls)) % len(palette)]])
    texwidths = resolutions(np.concatenate([agentlines, walls]))
    textures = core.gamma_decode(np.repeat(colors, texwidths, 0))
    pattern = wall_pattern(textures.shape[0], random=random)
    pattern[:texwidths[:len(agentlines)].sum()] = 1.
    return textures*pattern[:, None], texwidths


def random_lights(lights, random=np.random):
    """Appends a U(.5, 2) intensity column to (I, 2) light positions (reference: scene.py:70-73)."""
    return np.concatenate([lights, random.uniform(.5, 2., (len(lights), 1))], -1)


@torch.no_grad()
def scenery(geometries, n_agents=1, device='cuda', random=np.random, bake=True):
    """Packs a list of geometries into a :class:`~megastep_amd.cuda.Scenery` on ``device`` and bakes its lighting
    (reference: scene.py:75-100). ``bake=False`` skips the GPU bake, for host-only plumbing."""
    agentlines = np.tile(agent_model(), (n_agents, 1, 1))
    agentcolors = np.tile(agent_colors(), (n_agents, 1))

    lights, lines, textures = [], [], []
    for g in geometries:
        lights.append(random_lights(g['lights']))      # global np.random, as in the reference (scene.py:82)
        lines.append(np.concatenate([agentlines, g['walls']]))
        textures.append(init_textures(agentlines, agentcolors, g['walls'], random))

    def pack(vals, widths):
        vals = arrdict.torchify(np.concatenate(vals)).to(device)
        widths = arrdict.torchify(np.asarray(widths).reshape(-1)).to(device)
        return ragged.Ragged(vals.contiguous(), widths.contiguous())

    result = cuda.Scenery(
        n_agents=n_agents,
        lights=pack(lights, [len(l) for l in lights]),
        lines=pack(lines, [len(l) for l in lines]),
        textures=pack([t for t, _ in textures], np.concatenate([w for _, w in textures])),
        model=arrdict.torchify(agent_model()).to(device))
    if bake:
        cuda.bake(result)
    return result
