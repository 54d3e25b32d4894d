"""Single-room toy geometries (reference: megastep/toys.py:5-29)."""
import numpy as np
from . import geometry, arrdict


def _square(side, centre):
    angles = np.arange(np.pi/4, 2*np.pi, np.pi/2)
    return side/2**.5*np.stack([np.cos(angles), np.sin(angles)], -1) + centre


def box(width=5):
    """A box with one room and one light in the middle of it."""
    centre = width/2 + geometry.MARGIN
    corners = _square(width, centre)
    walls = np.stack(geometry.cyclic_pairs(corners))
    return arrdict.arrdict(
        walls=walls,
        lights=np.full((1, 2), centre),
        masks=geometry.masks(walls, [corners]),
        res=geometry.RES)


def column(width=5, column_width=.1):
    """A small square column with one big room around it, lit from the room's corners."""
    centre = width/2 + geometry.MARGIN
    inner, outer = _square(column_width, centre), _square(width, centre)
    walls = np.stack(geometry.cyclic_pairs(inner))
    return arrdict.arrdict(
        walls=walls,
        lights=_square(2., centre),
        masks=geometry.masks(walls, [outer]),
        res=geometry.RES)
