"""Builds scratch/probe.so: render_kernel<1,1,*> with cycle counters per section written over out.distances
(lanes 0-7 of every ray group; lane 8 = (line, ray) pairs) - for tools/probe_run.py. Results are garbage by design."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(f'{root}/megastep_amd/csrc/megastep_hip.hip').read()
def rep(old, new):
    global src
    assert src.count(old) >= 1, old
    src = src.replace(old, new, 1)
rep("    Cand* const s_cand_w = reinterpret_cast<Cand*>(&s_raw[wave][0]);",
    "    long long T_[8] = {0,0,0,0,0,0,0,0}; long long t_ = clock64(); int Psum = 0;\n#define TICK(k) { const long long n_ = clock64(); T_[k] += n_ - t_; t_ = n_; }\n    Cand* const s_cand_w = reinterpret_cast<Cand*>(&s_raw[wave][0]);")
rep("        for (int c0 = 0; c0 < L; c0 += WAVE) {\n            int lo = 0, len = 0;\n            line_setup(c0, lo, len);\n            const int incl = wave_scan_add(len);",
    "        TICK(0)\n        for (int c0 = 0; c0 < L; c0 += WAVE) {\n            int lo = 0, len = 0;\n            line_setup(c0, lo, len);\n            TICK(1)\n            const int incl = wave_scan_add(len);")
rep("            n_pairs_total += P; n_windows += (P + WAVE - 1)/WAVE;\n            int carry = -1;", "            n_pairs_total += P; n_windows += (P + WAVE - 1)/WAVE;\n            int carry = -1;\n            TICK(2)\n            Psum += P;")
rep("                __builtin_amdgcn_wave_barrier();\n            }\n            __builtin_amdgcn_wave_barrier();\n        }\n        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");\n        const unsigned long long best",
    "                __builtin_amdgcn_wave_barrier();\n            }\n            __builtin_amdgcn_wave_barrier();\n            TICK(3)\n        }\n        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");\n        const unsigned long long best")
rep("    // ---- the winner's loc and dot, recomputed from the same inputs", "    TICK(4)\n    // ---- the winner's loc and dot, recomputed from the same inputs")
rep("    // ---- pass 3: shade (kernels.cu:407-450)\n    const bool is_hit", "    TICK(5)\n    // ---- pass 3: shade (kernels.cu:407-450)\n    const bool is_hit")
rep("    float s0 = 0.f, s1 = 0.f, s2 = 0.f;\n", "    TICK(6)\n    float s0 = 0.f, s1 = 0.f, s2 = 0.f;\n")
src = src.rstrip()
assert src.count("    // ---- pooled observations") == 1
src = src.replace("    // ---- pooled observations", """    TICK(7)
    {
        int v = Psum;
        #pragma unroll
        for (int k = 0; k < 8; k++) if (lane == k) v = (int)T_[k];
        if (lane < 9) reinterpret_cast<int*>(out.distances)[((size_t)n*A + a)*R + g*WAVE + lane] = v;
    }
    // ---- pooled observations""", 1)
os.makedirs(f'{root}/scratch', exist_ok=True)
open(f'{root}/scratch/probe.hip', 'w').write(src.replace('../../include/megastep_hip.h', 'megastep_hip.h'))
flags = '--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize'.split()
subprocess.check_call(['/opt/rocm/bin/hipcc', *flags, f'-I{root}/include', '-o', f'{root}/scratch/probe.so', f'{root}/scratch/probe.hip'])
print('built scratch/probe.so')
