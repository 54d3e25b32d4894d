import sys; sys.path.insert(0,'.')
import numpy as np, torch
from tests import test_gpu_parity as T
from megastep_amd import cuda
# re-run the adversarial builders with the instrumented library and report how many rays took the full fallback
import types
captured = {}
orig = T.util.assert_render_matches
def spy(c, r, ref, atol=1e-5):
    amb = r.distances[..., 0].cpu().numpy()
    captured.setdefault('amb', []).append(amb)
T.util.assert_render_matches = spy
T.test_hysteresis_band_adversarial(); a = captured['amb'][-1]
print('band test: waves', a.size, 'full-fallback rays per wave', a.flatten().astype(int).tolist())
T.test_agent_wedged_between_coincident_walls(); a = captured['amb'][-1]
print('wedged test: full-fallback rays per wave', a.flatten().astype(int).tolist())
