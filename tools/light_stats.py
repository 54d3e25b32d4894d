"""How the dynamic lighting's work splits up on the benchmark world: rays on agents, grid verdicts, candidate lists."""
import sys; sys.path.insert(0, '.')
import torch, numpy as np, bench
from megastep_amd import cuda, modules
N, A, R = 4096, 4, 64
core, _ = bench.build_world(N, A, R, 130., torch.device('cuda'), seed=1)
mover = modules.MomentumMovement(core)
class D: pass
for i in range(60):
    D.actions = torch.randint(0, 7, (N, A), device='cuda'); mover(D)
r = cuda.render(core.scenery, core.agents)
sc = core.scenery
idx = r.indices.reshape(N, A, R)
loc = r.locations.reshape(N, A, R)
AF = A*sc.model.shape[0]
dyn = (idx >= 0) & (idx < AF)
print('rays on agents: %.3f of rays; fans with any: %.3f' % (dyn.float().mean().item(), dyn.any(-1).float().mean().item()))
vals, starts, geom, cell, _, lists, pool = sc._lg[:7]
print('pool: %d of %d words used; cells %d' % (int(pool[0]), len(pool), len(vals)))
e, a, k = dyn.nonzero(as_tuple=True)
lines = sc.lines.vals[sc.lines.starts.long()[e] + idx[e, a, k].long()]
t = loc[e, a, k][:, None]
C = lines[:, 0]*(1 - t) + lines[:, 1]*t
g = geom[e]
fx = torch.floor((C[:, 0] - g[:, 0])/cell).long(); fy = torch.floor((C[:, 1] - g[:, 1])/cell).long()
cid = starts.long()[e] + fy*g[:, 2].long() + fx
rec = torch.cat([vals[cid], lists[cid]], 1).long() & 0xffffffff
ni = sc.lights.widths.long()[e]
print('lights per env: mean %.1f max %d' % (sc.lights.widths.float().mean().item(), sc.lights.widths.max().item()))
w = rec[:, :4]
bits = torch.stack([(w[:, i >> 4] >> (2*(i & 15))) & 3 for i in range(64)], 1)
valid = torch.arange(64, device=bits.device)[None] < ni[:, None]
lit, dark, unk = ((bits == 1) & valid), ((bits == 2) & valid), ((bits == 0) & valid)
print('per ray: lit %.2f dark %.2f unknown %.2f' % (lit.sum(1).float().mean().item(), dark.sum(1).float().mean().item(), unk.sum(1).float().mean().item()))
L = sc.lights.vals[(sc.lights.starts.long()[e][:, None] + torch.arange(64, device=e.device)[None]).clamp(max=sc.lights.vals.shape[0] - 1)]
d2 = ((L[..., :2] - C[:, None])**2).sum(-1).clamp(min=1)
part = .1 + (2*L[..., 2]/d2*lit).sum(1)
sat = part >= 1.001
need = ~sat & unk.any(1)
print('rays: saturated %.3f, need walls %.3f (of agent rays)' % (sat.float().mean().item(), need.float().mean().item()))
listed = rec[:, -1] != 0
print('need rays with a list: %.3f; candidates per listed need ray: %.2f' % ((listed & need).float().sum().item()/max(need.sum().item(), 1),
      (rec[:, -1] & 0x7fffffff)[listed & need].float().mean().item()))
fan = (e*A + a)
nf = torch.zeros(N*A, device=e.device); nf.index_add_(0, fan[need], torch.ones(int(need.sum()), device=e.device))
sw = torch.zeros(N*A, device=e.device); sw.index_add_(0, fan[need & ~listed], torch.ones(int((need & ~listed).sum()), device=e.device))
print('candidates per listed need ray: quantiles', torch.quantile((rec[:, -1] & 0x7fffffff)[listed & need].float(), torch.tensor([.5, .9, .99, 1.], device=e.device)).tolist())
print('fans with need rays: %.3f; fans with sweep rays: %.3f' % ((nf > 0).float().mean().item(), (sw > 0).float().mean().item()))
