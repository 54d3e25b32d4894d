"""Per-section cycle counters (s_memtime) of render_kernel<2,1,*>: builds scratch/probe2.so with the counters written over
out.distances (lanes 0-7 of every ray group = sections, lanes 8-11 = pairs, windows, chunks with visible lines, drains),
runs it on the benchmark world and prints the breakdown. Results of the probed render are garbage by design.
usage: python tools/probe_v2.py [build|run|both]"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build():
    src = open(f'{root}/megastep_amd/csrc/megastep_hip.hip').read()
    i2 = src.index("    } else if constexpr (IMPL == 2) {")

    def rep(old, new, start=0):
        nonlocal src
        k = src.index(old, start)
        src = src[:k] + new + src[k + len(old):]
    rep("    Cand* const s_cand_w = reinterpret_cast<Cand*>(&s_raw[wave][0]);",
        "    long long T_[8] = {0,0,0,0,0,0,0,0}; long long t_ = clock64(); int n_vis_ = 0, n_drain_ = 0;\n#define TICK(k) { const long long n_ = clock64(); T_[k] += n_ - t_; t_ = n_; }\n    Cand* const s_cand_w = reinterpret_cast<Cand*>(&s_raw[wave][0]);")
    i2 = src.index("    } else if constexpr (IMPL == 2) {")
    rep("        auto drain = [&]() {\n", "        auto drain = [&]() {\n            TICK(4) n_drain_++;\n", i2)
    rep("            // the list starts over\n", "            TICK(3)\n            // the list starts over\n", i2)
    rep("        float4 w_next[AHEAD];", "        TICK(0)\n        float4 w_next[AHEAD];", i2)
    rep("            line_math(w, l, live, agent_lines, first_chunk, cd, lo, len);\n", "            TICK(1)\n            line_math(w, l, live, agent_lines, first_chunk, cd, lo, len);\n            TICK(2)\n", i2)
    rep("            if (!vm) return;                                                 // uniform\n", "            if (!vm) return;                                                 // uniform\n            n_vis_++;\n", i2)
    rep("        if (n_pairs) drain();\n", "        TICK(1)\n        if (n_pairs) drain();\n", i2)
    rep("    // ---- the winner's loc and dot, recomputed from the same inputs", "    TICK(5)\n    // ---- the winner's loc and dot, recomputed from the same inputs")
    rep("    // ---- pass 3: shade (kernels.cu:407-450)\n    const bool is_hit", "    TICK(6)\n    // ---- pass 3: shade (kernels.cu:407-450)\n    const bool is_hit")
    rep("    float s0 = 0.f, s1 = 0.f, s2 = 0.f;\n", "    TICK(7)\n    float s0 = 0.f, s1 = 0.f, s2 = 0.f;\n")
    # telemetry variables of IMPL 2 are local to its branch: export them through function-scope shadows
    rep("    int nearest_idx = -1;\n", "    int nearest_idx = -1; int probe_pairs_ = 0, probe_windows_ = 0;\n")
    i2 = src.index("    } else if constexpr (IMPL == 2) {")
    rep("        // The literal fold for the rays that need it (kernels.cu:352-377), as in IMPL 1\n", "        probe_pairs_ = n_pairs_total; probe_windows_ = n_windows;\n        // The literal fold for the rays that need it (kernels.cu:352-377), as in IMPL 1\n", i2)
    assert src.count("    // ---- pooled observations") == 1
    src = src.replace("    // ---- pooled observations", """    TICK(7)
    {
        int v = 0;
        #pragma unroll
        for (int k = 0; k < 8; k++) if (lane == k) v = (int)T_[k];
        if (lane == 8) v = probe_pairs_; if (lane == 9) v = probe_windows_; if (lane == 10) v = n_vis_; if (lane == 11) v = n_drain_;
        if (lane < 12) reinterpret_cast<int*>(out.distances)[((size_t)n*A + a)*R + g*WAVE + lane] = v;
    }
    // ---- pooled observations""", 1)
    os.makedirs(f'{root}/scratch', exist_ok=True)
    open(f'{root}/scratch/probe2.hip', 'w').write(src.replace('../../include/megastep_hip.h', 'megastep_hip.h'))
    flags = '--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize'.split()
    subprocess.check_call(['/opt/rocm/bin/hipcc', *flags, f'-I{root}/include', '-o', f'{root}/scratch/probe2.so', f'{root}/scratch/probe2.hip'])
    print('built scratch/probe2.so')


def run():
    os.environ['MEGASTEP_HIP_LIB'] = f'{root}/scratch/probe2.so'
    os.environ['MEGASTEP_RENDER_IMPL'] = 'v2'
    sys.path.insert(0, root)
    import numpy as np, torch, bench
    from megastep_amd import cuda, modules
    N, A, R = 4096, 4, 64
    core, _ = bench.build_world(N, A, R, 130., torch.device('cuda'), seed=1)
    mover = modules.MomentumMovement(core)

    class D:
        pass
    acc, cnt = np.zeros(12), 0
    for i in range(60):
        D.actions = torch.randint(0, 7, (N, A), device='cuda')
        mover(D)
        r = cuda.render(core.scenery, core.agents)
        if i >= 40:
            d = r.distances.reshape(N*A, R)[:, :12].contiguous().view(torch.int32).double()
            acc += d.mean(0).cpu().numpy(); cnt += 1
            if i == 59:
                tot = d[:, :8].sum(1)
                print('per-wave total cycles: mean %.0f  p50 %.0f p90 %.0f p99 %.0f max %.0f' % (tot.mean().item(), *[torch.quantile(tot, q).item() for q in (.5, .9, .99)], tot.max().item()))
                lines = core.scenery.lines.widths.repeat_interleave(A).double()
                slow = tot >= torch.quantile(tot, .99)
                print('slowest 1%% of the waves: mean sections', d[slow].mean(0).cpu().numpy().round(0).tolist(), 'lines/env %.0f vs %.0f overall' % (lines[slow].mean().item(), lines.mean().item()))
                worst = tot.argmax()
                print('the slowest wave:', d[worst].cpu().numpy().round(0).tolist(), 'lines', lines[worst].item())
                light = d[:, 7]
                print('dynamic lighting per wave: share of waves above 2k / 5k / 10k / 20k cycles: %s; their share of all wave cycles: %s' % (
                    [round((light > t).double().mean().item(), 4) for t in (2e3, 5e3, 1e4, 2e4)],
                    [round((tot[light > t].sum()/tot.sum()).item(), 4) for t in (2e3, 5e3, 1e4, 2e4)]))
                rest = tot - light
                print('without the lighting section: mean %.0f p99 %.0f max %.0f' % (rest.mean().item(), torch.quantile(rest, .99).item(), rest.max().item()))
                print('corr(total, lines) %.2f  corr(total, pairs) %.2f' % (torch.corrcoef(torch.stack([tot, lines]))[0, 1].item(), torch.corrcoef(torch.stack([tot, d[:, 8]]))[0, 1].item()))
    acc /= cnt
    names = ['prologue', 'line loads + loop', 'pass 1 line math', 'pass2 windows', 'scan+compaction', 'resolve+fold', 'loc/dot+out', 'lighting+shade+store',
             'pairs', 'windows', 'batches with visible lines', 'drains']
    tot = acc[:8].sum()
    for n, v in zip(names, acc):
        print('%-26s %10.1f  %s' % (n, v, '%.1f%%' % (100*v/tot) if names.index(n) < 8 else ''))


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'both'
    if what in ('build', 'both'):
        build()
    if what in ('run', 'both'):
        run()
