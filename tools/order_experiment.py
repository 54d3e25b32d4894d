"""What would taking a render launch's fans longest-first buy?  (-DMS_ORDER_EXPERIMENT builds: tools/build_variants.sh
"order:-DMS_ORDER_EXPERIMENT=1" "order_cost:-DMS_ORDER_EXPERIMENT=2")

On the benchmark world after a few steps: one launch of the cost build records every wave's life; orders are made from
those on the host - identity, each XCD run sorted by life (longest first: the ideal), two buckets at a threshold laid out
the way a kernel could build them (segments of a run, slow from the front, fast from the back, segments interleaved) - and
the order build is timed with each, next to the product library.

    python tools/order_experiment.py [--envs 4096 --agents 4 --res 64]"""
import argparse, ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np, torch, bench                                        # noqa: E402
from megastep_amd import _lib, cuda, modules                             # noqa: E402


def load(path):
    _lib._lib = None
    _lib.LIB_PATH = path
    os.environ['MEGASTEP_HIP_LIB'] = path
    return _lib.lib()


def runs_of(n_fans):
    q8, r8 = n_fans >> 3, n_fans & 7
    starts = [x*q8 + min(x, r8) for x in range(9)]
    return list(zip(starts[:-1], starts[1:]))


def sorted_order(cost):
    order = np.arange(len(cost), dtype=np.int32)
    for a, b in runs_of(len(cost)):
        order[a:b] = a + np.argsort(-cost[a:b], kind='stable')
    return order


def bucket_order(cost, frac, seg=256):
    """Two buckets per segment of a run, slow from the front and fast from the back, segments interleaved."""
    theta = np.quantile(cost, 1 - frac)
    order = np.arange(len(cost), dtype=np.int32)
    for a, b in runs_of(len(cost)):
        segs = [np.arange(s, min(s + seg, b)) for s in range(a, b, seg)]
        placed = []
        for fans in segs:
            slow = fans[cost[fans] > theta]
            fast = fans[cost[fans] <= theta]
            placed.append(np.concatenate([slow, fast[::-1]]))
        out, p = [], 0
        while len(out) < b - a:
            for s in placed:
                if p < len(s):
                    out.append(s[p])
            p += 1
        order[a:b] = np.array(out, dtype=np.int32)
    return order


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=4096); ap.add_argument('--agents', type=int, default=4)
    ap.add_argument('--res', type=int, default=64); ap.add_argument('--large', action='store_true')
    ap.add_argument('--unique', type=int, default=512); ap.add_argument('--fast-build', action='store_true')
    args = ap.parse_args()
    v = f'{root}/megastep_amd/csrc/variants'
    h_prod = load(f'{root}/megastep_amd/csrc/libmegastep_hip.so')
    core, _ = bench.build_world(args.envs, args.agents, args.res, 130., torch.device('cuda'), seed=1, n_unique=args.unique,
                                large=args.large, fast=args.fast_build)
    N, A = core.n_envs, core.n_agents
    n_fans = N*A*((args.res + 63)//64)
    mover = modules.MomentumMovement(core)
    for i in range(12):
        actions = torch.randint(0, 7, (N, A), device='cuda')
        delta = mover._actionset[actions]
        core.agents.angvelocity[:] = .875*core.agents.angvelocity + delta.angvelocity
        core.agents.velocity[:] = .875*core.agents.velocity + modules.to_global_frame(core.agents.angles, delta.velocity)
        cuda.physics(core.scenery, core.agents)
        ref = cuda.render(core.scenery, core.agents)
    torch.cuda.synchronize()
    want = {k: getattr(ref, k).clone() for k in ('indices', 'screen', 'distances')}

    def timed(h, order, launches=20, reps=7):
        _lib._lib = h
        if order is not None:
            _lib.check(h.ms_debug_order(order.data_ptr(), None))
        for _ in range(3):
            out = cuda.render(core.scenery, core.agents)
        torch.cuda.synchronize()
        same = all(torch.equal(getattr(out, k), want[k]) for k in want)
        g = torch.cuda.CUDAGraph()                      # (back to back on the GPU: an eager launch's events also time the host)
        with torch.cuda.graph(g):
            for _ in range(launches):
                out = cuda.render(core.scenery, core.agents)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ts = []
        for _ in range(reps + 2):
            ev[0].record(); g.replay(); ev[1].record()
            torch.cuda.synchronize()
            ts.append(1e3*ev[0].elapsed_time(ev[1])/launches)
        return np.median(ts[2:]), np.min(ts[2:]), same

    identity = torch.arange(n_fans, dtype=torch.int32, device='cuda')
    h_cost = load(f'{v}/order_cost.so')
    h_cost.ms_debug_order.argtypes = [C.c_void_p, C.c_void_p]
    cost_t = torch.zeros(n_fans, dtype=torch.int32, device='cuda')
    _lib.check(h_cost.ms_debug_order(identity.data_ptr(), cost_t.data_ptr()))
    costs = []
    for _ in range(5):
        cuda.render(core.scenery, core.agents)
        torch.cuda.synchronize()
        costs.append(cost_t.cpu().numpy().view(np.uint32).astype(np.float64))
    _lib.check(h_cost.ms_debug_order(identity.data_ptr(), None))
    cost = np.median(costs, 0)
    print(f'lives (shader clocks): mean {cost.mean():.0f} p50 {np.median(cost):.0f} p90 {np.quantile(cost, .9):.0f} p99 {np.quantile(cost, .99):.0f} max {cost.max():.0f}; '
          f'launch to launch, the same fan: {np.mean(np.abs(costs[0] - costs[1])/cost):.3f} relative', flush=True)

    # one step on: how much of it survives when the lives are those of the step before?
    def record():
        _lib._lib = h_cost
        _lib.check(h_cost.ms_debug_order(identity.data_ptr(), cost_t.data_ptr()))
        cs = []
        for _ in range(5):
            cuda.render(core.scenery, core.agents)
            torch.cuda.synchronize()
            cs.append(cost_t.cpu().numpy().view(np.uint32).astype(np.float64))
        _lib.check(h_cost.ms_debug_order(identity.data_ptr(), None))
        return np.median(cs, 0)
    old_cost = cost
    _lib._lib = h_prod
    actions = torch.randint(0, 7, (N, A), device='cuda')
    delta = mover._actionset[actions]
    core.agents.angvelocity[:] = .875*core.agents.angvelocity + delta.angvelocity
    core.agents.velocity[:] = .875*core.agents.velocity + modules.to_global_frame(core.agents.angles, delta.velocity)
    cuda.physics(core.scenery, core.agents)
    ref = cuda.render(core.scenery, core.agents)
    torch.cuda.synchronize()
    want = {k: getattr(ref, k).clone() for k in ('indices', 'screen', 'distances')}
    cost = record()
    F_ = n_fans//N
    slow_old = old_cost.reshape(N, F_).max(1) > np.quantile(old_cost.reshape(N, F_).max(1), .8)
    slow_new = cost.reshape(N, F_).max(1) > np.quantile(cost.reshape(N, F_).max(1), .8)
    print(f'envs among the slowest fifth in both steps: {np.mean(slow_old & slow_new)/np.mean(slow_new):.2f} of them', flush=True)
    h_ord = load(f'{v}/order.so')
    h_ord.ms_debug_order.argtypes = [C.c_void_p, C.c_void_p]
    F = n_fans//N
    env_cost = np.repeat(cost.reshape(N, F).max(1), F)                     # an env's fans stay together: its slowest fan's life
    orders = {'product library': None, 'identity': identity.cpu().numpy(), 'sorted per run (ideal)': sorted_order(cost),
              'envs sorted per run': sorted_order(env_cost)}
    for frac in (.05, .1, .2):
        orders[f'two buckets, slowest {frac:.2f}'] = bucket_order(cost, frac)
        orders[f'two buckets of envs, {frac:.2f}'] = bucket_order(env_cost, frac)
    # a stale predictor: 15 % of the fans' costs shuffled
    rng = np.random.default_rng(0)
    stale = cost.copy(); idx = rng.choice(n_fans, n_fans*15//100, replace=False); stale[idx] = stale[rng.permutation(idx)]
    orders['sorted, 15 % of costs wrong'] = sorted_order(stale)
    old_env_cost = np.repeat(old_cost.reshape(N, F).max(1), F)
    orders['envs sorted, lives one step old'] = sorted_order(old_env_cost)
    orders['two buckets of envs 0.20, one step old'] = bucket_order(old_env_cost, .2)
    orders['two buckets 0.10, one step old'] = bucket_order(old_cost, .1)
    results = {k: [] for k in orders}
    for cycle in range(4):                                                 # (interleaved: the box's clocks drift over a run)
        for name, o in orders.items():
            if o is None:
                t = timed(h_prod, None)
            else:
                assert sorted(o.tolist()) == list(range(n_fans))
                t = timed(h_ord, torch.as_tensor(o, device='cuda'))
            assert t[2]
            if cycle: results[name].append(t[0])
    for name, ts in results.items():
        print('%-40s: %s us per launch (median %.1f)' % (name, ' '.join('%.1f' % t for t in ts), np.median(ts)), flush=True)
    print('done', flush=True)


if __name__ == '__main__':
    main()
