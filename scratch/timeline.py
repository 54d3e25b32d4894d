import sys; sys.path.insert(0,'.')
import torch, numpy as np
import bench
from megastep_amd import cuda, modules
A = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = 16384//A
core,_ = bench.build_world(N, A, 64, 130., torch.device('cuda'), seed=1)
mover = modules.MomentumMovement(core)
for i in range(30):
    class D: actions = torch.randint(0,7,(N,A),device='cuda')
    mover(D)
for _ in range(3): r = cuda.render(core.scenery, core.agents)
st = r.locations[...,0].flatten().cpu().numpy().astype(np.int64); en = r.dots[...,0].flatten().cpu().numpy().astype(np.int64)
# unwrap 24-bit
t0 = st.min(); 
st = (st - t0) % (1<<24); en = (en - t0) % (1<<24)
dur = (en - st) % (1<<24)
print('waves', len(st), 'kernel span cycles', en.max(), 'wave duration quantiles', np.quantile(dur, [.1,.5,.9,.99,1]).round())
# concurrency over time
edges = np.linspace(0, en.max(), 21)
for a,b in zip(edges[:-1], edges[1:]):
    mid = (a+b)/2
    print(f'{mid:9.0f}: running {((st<=mid)&(en>mid)).sum():5d}  started {((st>=a)&(st<b)).sum():5d}')
