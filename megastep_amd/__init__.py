"""megastep_amd: an MI355X-native (gfx950) simulation core with megastep's operator surface.

``megastep_amd.cuda`` is the drop-in for the reference's ``megastep.cuda`` extension module (reference:
megastep/__init__.py:7-20): same names, same tensors, hand-written HIP kernels behind a C-ABI (include/megastep_hip.h).
``core``, ``ragged``, ``scene``, ``modules``, ``toys``, ``spaces``, ``geometry`` mirror the host-side modules that call
it; ``cubicasa`` is a seeded synthetic stand-in for the (network-fetched) dataset.
"""
DEBUG = False

from . import dotdict, arrdict, cuda, ragged, spaces, geometry, core, scene, toys, modules, cubicasa, sharding  # noqa: E402,F401
