"""How a launch of physics_kernel / render_kernel fills and drains the chip: every wave's time stamps from the probe
build (`make -C megastep_amd/csrc probe`; s_memtime ticks once per shader clock, 2.4 GHz on MI355X), on the benchmark world.

    python tools/probe_waves.py [--envs 4096 --agents 4 --res 64 [--large --unique 64]]

Prints, per kernel: the launch's span, when waves start (the dispatch ramp) and how long they live, how many are
resident over time, and the mean time between a wave's stamps (what it waited for)."""
import argparse, ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ['MEGASTEP_HIP_LIB'] = f'{root}/megastep_amd/csrc/libmegastep_hip_probe.so'
import numpy as np, torch, bench                                        # noqa: E402
from megastep_amd import _lib, cuda, modules                             # noqa: E402

WORDS, STAMPS = 16, 8
NAMES = {'physics': ['start', 'agent state in', 'cell headers in', 'walls met', '', '', '', 'end'],
         'render': ['start', 'agent state in', '', 'first rows in', 'raycast done', 'winner row + texel row in', 'texels in', 'end']}


def analyse(name, rec, tick_ns, kernel_us):
    rec = rec[rec[:, STAMPS - 1] != 0]                                   # (waves that wrote a record)
    where = rec[:, STAMPS]
    xcc = (where >> 16) & 0xf
    t = rec[:, :STAMPS].astype(np.int64)
    # s_memtime (the stamps) counts per XCD: sections and lives come from it; the launch's time line from the 100 MHz
    # real-time counter the waves read at their start and end (32-bit words: differences modulo 2^32)
    t = ((t - t[:, :1]).astype(np.int32))*tick_ns/1e3
    r0, r1 = rec[:, STAMPS + 1].astype(np.int64), rec[:, STAMPS + 2].astype(np.int64)
    base = r0[np.argmin((r0 - r0[0]).astype(np.int32))]
    start, end = (r0 - base).astype(np.int32)/100., (r1 - base).astype(np.int32)/100.
    life = t[:, STAMPS - 1]
    print(f'== {name}: kernel {kernel_us:.1f} us by events; {len(rec)} waves, span {end.max():.1f} us; starts: p50 {np.median(start):.1f} p90 {np.quantile(start, .9):.1f} '
          f'last {start.max():.1f} us; life: mean {life.mean():.2f} p50 {np.median(life):.2f} p90 {np.quantile(life, .9):.2f} p99 {np.quantile(life, .99):.2f} max {life.max():.2f} us')
    edges = np.linspace(0, end.max(), 25)[:-1]
    print(f'   resident waves every {edges[1]:.2f} us:', [(int(((start <= a) & (end > a)).sum())) for a in edges])
    print(f'   per XCD: last end', [round(float(end[xcc == x].max()), 1) for x in np.unique(xcc)], 'waves', [int((xcc == x).sum()) for x in np.unique(xcc)])
    used = [i for i in range(STAMPS) if NAMES[name][i]]
    for a, b in zip(used[:-1], used[1:]):
        d = t[:, b] - t[:, a]
        print(f'   {NAMES[name][a]:>26s} -> {NAMES[name][b]:<26s} mean {d.mean():6.2f}  p50 {np.median(d):6.2f}  p90 {np.quantile(d, .9):6.2f} us')
    # how waves fare by when they start (a section that shrinks as the chip empties is contention, one that does not is latency)
    qs = np.quantile(start, np.linspace(0, 1, 9))
    print('   by start time (eighths of the waves): ' + ' | '.join(f'{qs[i]:.0f}-{qs[i + 1]:.0f} us: life {life[(start >= qs[i]) & (start <= qs[i + 1])].mean():.1f}' for i in range(8)))
    for a, b in zip(used[:-1], used[1:]):
        d = t[:, b] - t[:, a]
        print(f'   {NAMES[name][a]:>26s} -> {NAMES[name][b]:<26s} ' + ' '.join(f'{d[(start >= qs[i]) & (start <= qs[i + 1])].mean():5.2f}' for i in range(8)))
    last = np.argsort(-end)[:12]
    print('   the 12 waves that end last (start, life | sections):', [(round(float(start[i]), 1), round(float(life[i]), 1), [round(float(t[i, b] - t[i, a]), 1) for a, b in zip(used[:-1], used[1:])]) for i in last])
    simd = xcc*65536 + (where & 0xfff0)                                   # xcc | se, sh, cu, simd (wave slot masked off)
    per = np.unique(simd, return_counts=True)[1]
    print(f'   SIMDs used {len(per)}; waves per SIMD: min {per.min()} mean {per.mean():.1f} max {per.max()}')
    if name == 'physics':                                                  # slots 4..6: pairs dealt, swept instead, agents stopped
        extra = rec[:, 4:7]
        order = np.argsort(-life)
        print('   (pairs, swept, stopped, life) of the 12 slowest waves:', [tuple(int(v) for v in extra[i]) + (round(float(life[i]), 1),) for i in order[:12]])
        for lo_, hi_ in ((0, 1), (1, 2), (2, 9)):
            m = (extra[:, 2] >= lo_) & (extra[:, 2] < hi_)
            if m.any():
                print(f'   waves with {lo_}..{hi_ - 1} stopped agents: {int(m.sum())}, life mean {life[m].mean():.2f} p99 {np.quantile(life[m], .99):.2f}; pairs mean {extra[m, 0].mean():.1f} max {extra[m, 0].max()}; swept {int(extra[m, 1].sum())}')
    if name == 'render':                       # slot 2: 0 no ray on an agent; bit 31 some, settled by the grid's verdicts; else the work left
        v = rec[:, 2]
        light = t[:, 5] - t[:, 4] if NAMES['render'][5] else 0.*t[:, 4]
        dyn, need = v != 0, (v != 0) & (v != 0x80000000)
        print(f'   waves with a ray on an agent: {int(dyn.sum())} ({light[dyn & ~need].mean():.2f} us from raycast to winner row, others {light[~dyn].mean():.2f}); '
              f'with open lights: {int(need.sum())} ({light[need].mean():.2f} us; rays {(v[need] & 127).mean():.1f}, of them without a list {((v[need] >> 7) & 127).mean():.2f}, '
              f'lists {((v[need] >> 14) & 15).mean():.2f}, rounds of pairs {((v[need] >> 18) & 127).mean():.2f}, lights {(v[need] >> 25).mean():.1f})')
        last = np.argsort(-end)[:12]
        print('   the 12 waves that end last (open rays, without list, lists, rounds, lights):', [(int(x & 127), int((x >> 7) & 127), int((x >> 14) & 15), int((x >> 18) & 127), int(x >> 25)) for x in v[last]])
    if name == 'render':                       # words 11..13: pairs, length of the list if it is whole in LDS (-1: not), rays re-done by the literal fold; 15: clock before that fold
        x = rec[:, 11:14].astype(np.int64)
        x[:, 1] = x[:, 1].astype(np.int32)
        raw = rec[:, :16].astype(np.int64)
        pass12 = ((raw[:, 15] - raw[:, 3]).astype(np.int32))*tick_ns/1e3
        fold = ((raw[:, 4] - raw[:, 15]).astype(np.int32))*tick_ns/1e3
        order = np.argsort(-life)
        print('   the 16 longest-lived waves (life, passes 1 + 2, fold us | pairs, list, folded rays):', [(round(float(life[i]), 1), round(float(pass12[i]), 1), round(float(fold[i]), 1)) + tuple(int(v) for v in x[i]) for i in order[:16]])
        print(f'   all waves: pairs mean {x[:, 0].mean():.0f} p99 {np.quantile(x[:, 0], .99):.0f} max {x[:, 0].max()}; list mean {x[:, 1].mean():.0f} max {x[:, 1].max()}, not whole {int((x[:, 1] < 0).sum())}; '
              f'passes 1 + 2 mean {pass12.mean():.2f} p99 {np.quantile(pass12, .99):.2f} us')
        for lo_, hi_ in ((0, 1), (1, 2), (2, 7), (7, 65)):
            mm = (x[:, 2] >= lo_) & (x[:, 2] < hi_)
            if mm.any():
                print(f'   waves with {lo_}..{hi_ - 1} folded rays: {int(mm.sum())}: life mean {life[mm].mean():.2f} p90 {np.quantile(life[mm], .9):.2f} max {life[mm].max():.2f} us; '
                      f'passes 1 + 2 mean {pass12[mm].mean():.2f} max {pass12[mm].max():.2f}; resolution + fold mean {fold[mm].mean():.2f} max {fold[mm].max():.2f} us')
    if name == 'render' and NAMES['render'][5]:
        # waves with rays to light: words 14, 12, 13, 11 = clocks at the lighting's start, when the lights' rows and the cell's
        # verdicts are in, after the sum over the LIT lights, at its end (they overwrite the pair statistics of those waves)
        v = rec[:, 2]
        dyn, need = v != 0, (v != 0) & (v != 0x80000000)
        raw = rec[:, :16].astype(np.int64)
        d = lambda a, b: ((raw[:, a] - raw[:, b]).astype(np.int32))*tick_ns/1e3
        for label, m in (('settled by the grid', dyn & ~need), ('with open lights', need)):
            if m.any():
                print(f'   lighting, waves {label} ({int(m.sum())}): raycast done -> start {d(14, 4)[m].mean():.2f}, -> rows in {d(12, 14)[m].mean():.2f}, '
                      f'-> LIT sum done {d(13, 12)[m].mean():.2f}, -> end {d(11, 13)[m].mean():.2f}, -> texel row used {d(5, 11)[m].mean():.2f} us; life {life[m].mean():.2f} (others {life[~dyn].mean():.2f})')
    slowest = np.argsort(-life)[:len(life)//100 + 1]
    print(f'   the slowest 1 %: starts at {np.median(start[slowest]):.1f} us (median), ' + ', '.join(
        f'{NAMES[name][a]}->{NAMES[name][b]} {np.mean(t[slowest, b] - t[slowest, a]):.2f}' for a, b in zip(used[:-1], used[1:])))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=4096); ap.add_argument('--agents', type=int, default=4)
    ap.add_argument('--res', type=int, default=64); ap.add_argument('--large', action='store_true')
    ap.add_argument('--unique', type=int, default=512); ap.add_argument('--fast-build', action='store_true')
    ap.add_argument('--tick-ns', type=float, default=1/2.4, help='one s_memtime tick in ns')
    ap.add_argument('--fov', type=float, default=130.); ap.add_argument('--depth-only', action='store_true')
    args = ap.parse_args()
    if args.depth_only:                                                   # (no shading pass: those stamps are never taken)
        NAMES['render'][5] = NAMES['render'][6] = ''
    h = _lib.lib()
    h.ms_debug_probe.argtypes = [C.c_void_p, C.c_longlong]
    core, _ = bench.build_world(args.envs, args.agents, args.res, args.fov, torch.device('cuda'), seed=1, n_unique=args.unique,
                                large=args.large, fast=args.fast_build)
    N, A = core.n_envs, core.n_agents
    mover = modules.MomentumMovement(core)
    cap = N*A*((args.res + 63)//64)
    buf = torch.zeros(cap*WORDS, dtype=torch.int32, device='cuda')
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for i in range(12):
        actions = torch.randint(0, 7, (N, A), device='cuda')
        delta = mover._actionset[actions]
        core.agents.angvelocity[:] = .875*core.agents.angvelocity + delta.angvelocity
        core.agents.velocity[:] = .875*core.agents.velocity + modules.to_global_frame(core.agents.angles, delta.velocity)
        for name in ('physics', 'render'):
            record = i == 11
            torch.cuda.synchronize()
            if record:
                buf.zero_()
                torch.cuda.synchronize()
                _lib.check(h.ms_debug_probe(buf.data_ptr(), cap))
            ev[0].record()
            if name == 'physics':
                cuda.physics(core.scenery, core.agents)
            else:
                cuda.render(core.scenery, core.agents, fields=('distances',) if args.depth_only else None)
            ev[1].record()
            torch.cuda.synchronize()
            if record:
                _lib.check(h.ms_debug_probe(None, 0))
                n = N if name == 'physics' else cap
                rec = buf[:n*WORDS].view(n, WORDS).cpu().numpy().view(np.uint32).astype(np.int64)
                analyse(name, rec, args.tick_ns, 1e3*ev[0].elapsed_time(ev[1]))
    print('done', flush=True)


if __name__ == '__main__':
    main()
