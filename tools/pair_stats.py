"""Pairs, windows and folded rays per wave of the pair raycasts on the benchmark world (telemetry path of ms_render)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from megastep_amd import cuda
core, _ = bench.build_world(4096, 4, 64, 130., torch.device('cuda'), seed=1)
waves = core.n_envs*core.n_agents*((core.res + 63)//64)
impls = sys.argv[1:] or ['v2']
if any(i != 'v2' for i in impls) and 'libmegastep_hip_ab' not in os.environ.get('MEGASTEP_HIP_LIB', ''):
    # only the A/B build (make -C megastep_amd/csrc ab) holds the older raycasts and reads MEGASTEP_RENDER_IMPL
    raise SystemExit('pair_stats.py: MEGASTEP_RENDER_IMPL=pairs|seq needs MEGASTEP_HIP_LIB=.../libmegastep_hip_ab.so')
for impl in impls:
    os.environ['MEGASTEP_RENDER_IMPL'] = impl
    r = cuda.render(core.scenery, core.agents, telemetry=True)
    torch.cuda.synchronize()
    q, folded, lanepar, pairs, windows = r._telemetry[:5].tolist()
    print(f'{impl}: pairs/wave {pairs/waves:.1f}  windows/wave {windows/waves:.2f}  folded rays {folded} ({folded/waves:.4f}/wave)  lane-parallel waves {lanepar}')
