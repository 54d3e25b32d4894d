"""Random worlds against the oracle (tools/fuzz_parity.py): shapes the fixed-shape parity tests do not name - up to 70
agents per env, 1 to 600 rays, narrow and wide views, toys and floorplans, and for odd seeds walls moved onto each other,
collapsed to points or stretched a hundredfold - bake, then three physics + render steps each; and the render kernel's
optional outputs (pooled RGB-D, crosshair ids, first-sight books, subsets of the planes) against tensor ops on a full render."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location(
    'fuzz_parity', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'fuzz_parity.py'))
fuzz = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(fuzz)


@pytest.mark.parametrize('first', [0, 12, 24, 36])
def test_random_worlds_match_the_oracle(first):
    for seed in range(first, first + 12):
        print(seed, fuzz.one(seed))
        print(seed, fuzz.one_fused(seed))


@pytest.mark.parametrize('first', [100, 108])
def test_random_oblique_worlds_match_the_oracle(first):
    """Round 6: every seed on floorplans turned by seeded angles with diagonal partitions (cubicasa.sample(oblique=True)) - until
    then every plan-scale world the device had met was axis-aligned, which the reference's SVG polygons are not."""
    for seed in range(first, first + 8):
        print(seed, fuzz.one(seed, oblique=True))
