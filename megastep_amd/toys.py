"""Two hand-made geometries for tests, tutorials and the minimal env: an empty square room and a room with a thin
column in the middle. Same outputs as megastep/toys.py:5-29 (the box's walls and masks are pinned by goldens).

A geometry is a dict: ``walls`` (W, 2, 2) segments, ``lights`` (I, 2) positions, ``masks`` int grid (-1 wall, 0 free,
k > 0 room k) and its ``res`` in metres per cell."""
import numpy as np

from . import arrdict, geometry


def _diamond_to_square(diagonal_half, centre):
    """The four corners of an axis-aligned square, counter-clockwise from the top-right one, given half its diagonal:
    points at 45, 135, 225 and 315 degrees on a circle of that radius."""
    turns = np.pi/4 + np.pi/2*np.arange(4)
    return diagonal_half*np.column_stack([np.cos(turns), np.sin(turns)]) + centre


def _geometry(wall_corners, room_corners, lights):
    walls = np.stack(geometry.cyclic_pairs(wall_corners))
    return arrdict.arrdict(walls=walls, lights=lights, masks=geometry.masks(walls, [room_corners]), res=geometry.RES)


def box(width=5):
    """A ``width`` m square room, its walls ``geometry.MARGIN`` away from the axes, one light in the middle."""
    middle = width/2 + geometry.MARGIN
    corners = _diamond_to_square(width/2**.5, middle)
    return _geometry(corners, corners, np.full((1, 2), middle))


def column(width=5, column_width=.1):
    """A ``column_width`` m square column standing in the middle of a ``width`` m room (which has no outer walls of
    its own), lit from four points around it."""
    middle = width/2 + geometry.MARGIN
    return _geometry(wall_corners=_diamond_to_square(column_width/2**.5, middle),
                     room_corners=_diamond_to_square(width/2**.5, middle),
                     lights=_diamond_to_square(2**.5, middle))
