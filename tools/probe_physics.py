"""Per-section cycle counters (s_memtime) of physics_kernel<0,0>: builds scratch/probe_phys.so with the counters written
into the heading cache (16 ints per env at 4 agents), runs it on the benchmark world, prints the breakdown.
usage: python tools/probe_physics.py [build|run|both]"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build():
    src = open(f'{root}/megastep_amd/csrc/megastep_hip.hip').read()

    def rep(old, new):
        nonlocal src
        assert src.count(old) == 1, old
        src = src.replace(old, new)
    rep("    extern __shared__ float4 s_dyn[];",
        "    long long T_[8] = {0,0,0,0,0,0,0,0}; long long t_ = clock64(); int n_flush_ = 0;\n#define TICKP(k) { const long long n_ = clock64(); T_[k] += n_ - t_; t_ = n_; }\n    extern __shared__ float4 s_dyn[];")
    rep("    // the spawn pose of agent i, if it is to be respawned (modules.py:321-326)\n", "    TICKP(0)\n    // the spawn pose of agent i, if it is to be respawned (modules.py:321-326)\n")
    rep("    // ... and the agent-agent tests (kernels.cu:193-200), one ordered pair per lane\n", "    TICKP(1)\n    // ... and the agent-agent tests (kernels.cu:193-200), one ordered pair per lane\n")
    rep("    int cnt = 0;\n    auto flush = [&]() {\n", "    TICKP(2)\n    int cnt = 0;\n    auto flush = [&]() {\n        TICKP(3) n_flush_++;\n")
    rep("        __builtin_amdgcn_wave_barrier();\n        cnt = 0;\n    };\n    // lane = wall: which agents' reach boxes", "        __builtin_amdgcn_wave_barrier();\n        cnt = 0;\n        TICKP(4)\n    };\n    // lane = wall: which agents' reach boxes")
    rep("    if (cnt) flush();\n    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");\n    // epilogue, kernels.cu:224-227\n", "    TICKP(3)\n    if (cnt) flush();\n    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");\n    TICKP(5)\n    // epilogue, kernels.cu:224-227\n")
    rep("        progress[i] = x;\n", "        progress[i] = x;\n        if (t == 0 && ag.headings) { TICKP(6) for (int k = 0; k < 8; k++) reinterpret_cast<int*>(ag.headings)[16*n + k] = (int)T_[k]; reinterpret_cast<int*>(ag.headings)[16*n + 8] = n_flush_; }\n")
    os.makedirs(f'{root}/scratch', exist_ok=True)
    open(f'{root}/scratch/probe_phys.hip', 'w').write(src.replace('../../include/megastep_hip.h', 'megastep_hip.h'))
    flags = '--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize'.split()
    subprocess.check_call(['/opt/rocm/bin/hipcc', *flags, f'-I{root}/include', '-o', f'{root}/scratch/probe_phys.so', f'{root}/scratch/probe_phys.hip'])
    print('built scratch/probe_phys.so')


def run():
    os.environ['MEGASTEP_HIP_LIB'] = f'{root}/scratch/probe_phys.so'
    sys.path.insert(0, root)
    import numpy as np, torch, bench
    from megastep_amd import cuda, modules
    N, A, R = 4096, 4, 64
    core, _ = bench.build_world(N, A, R, 130., torch.device('cuda'), seed=1)
    mover = modules.MomentumMovement(core)
    acc, cnt = np.zeros(9), 0
    for i in range(40):
        actions = torch.randint(0, 7, (N, A), device='cuda')
        delta = mover._actionset[actions]
        core.agents.angvelocity[:] = .875*core.agents.angvelocity + delta.angvelocity
        core.agents.velocity[:] = .875*core.agents.velocity + modules.to_global_frame(core.agents.angles, delta.velocity)
        cuda.physics(core.scenery, core.agents)
        if i >= 20:
            d = core.agents._headings.view(torch.int32).reshape(N, 16)[:, :9].double()
            acc += d.mean(0).cpu().numpy(); cnt += 1
            tot = d[:, :7].sum(1)
    acc /= cnt
    print('per-wave total cycles: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f' % (tot.mean().item(), *[torch.quantile(tot, q).item() for q in (.5, .9, .99)], tot.max().item()))
    names = ['start: kernargs, env row, first wall requests', 'agent state + reach boxes', 'agent-agent tests', 'wall sweep (box tests)', 'flushes (exact tests)', 'final fence', 'epilogue', '-', 'flushes per env']
    total = acc[:7].sum()
    for n, v in zip(names, acc):
        print('%-48s %9.1f  %s' % (n, v, '%.1f%%' % (100*v/total) if names.index(n) < 7 else ''))


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'both'
    if what in ('build', 'both'):
        build()
    if what in ('run', 'both'):
        run()
