import sys; sys.path.insert(0,'.')
import torch, numpy as np, bench
from megastep_amd import cuda, modules
N,A,R=4096,4,64
core,_ = bench.build_world(N, A, R, 130., torch.device('cuda'), seed=1)
mover = modules.MomentumMovement(core)
class D: pass
acc=np.zeros(9); cnt=0
for i in range(60):
    D.actions = torch.randint(0,7,(N,A),device='cuda'); mover(D)
    r = cuda.render(core.scenery, core.agents)
    if i>=40:
        d = r.distances.reshape(N*A, R)[:, :9].contiguous().view(torch.int32).double()
        acc += d.mean(0).cpu().numpy(); cnt+=1
        if i==59:
            tot = d[:, :8].sum(1)
            print('per-wave total cycles: mean %.0f  p50 %.0f p90 %.0f p99 %.0f max %.0f' % (tot.mean().item(), *[torch.quantile(tot, q).item() for q in (.5,.9,.99)], tot.max().item()))
            for k in (1,3,7):
                print('section', k, 'p50 %.0f p90 %.0f p99 %.0f max %.0f' % (*[torch.quantile(d[:,k], q).item() for q in (.5,.9,.99)], d[:,k].max().item()))
acc/=cnt
names=['prologue','pass1 line_setup','cull+scan+info','pass2 pairs','resolve+fallback','loc/dot+out','lighting','shade+store','pairs']
tot=acc[:8].sum()
for n,v in zip(names,acc):
    print('%-18s %10.0f cyc/wave  %.1f%%' % (n, v, 100*v/tot))
