"""Static instruction counts per basic block of a kernel in the gfx950 assembly of megastep_hip.hip (hipcc -S): where a
wave's vector instructions are.  usage: python tools/static_profile.py [kernel-name-substring] [min VALU per block]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1] if len(sys.argv) > 1 else 'render_kernelILi2ELi1ELi0E'
floor = int(sys.argv[2]) if len(sys.argv) > 2 else 20
flags = '--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize --cuda-device-only -S'.split()
subprocess.run(['/opt/rocm/bin/hipcc', *flags, *sys.argv[3:], '-o', '/tmp/ms.s', f'{root}/megastep_amd/csrc/megastep_hip.hip'], check=True, stderr=subprocess.DEVNULL)
lines = open('/tmp/ms.s').read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and want in l and ': ' in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.section') or lines[i].strip().startswith('.end_amdhsa_kernel') or '.Lfunc_end' in lines[i])
blocks, cur = [], ['entry', []]
for l in lines[start + 1:end]:
    t = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', t):
        blocks.append(cur); cur = [t.rstrip(':'), []]
    elif t and not t.startswith((';', '.')):
        cur[1].append(t)
blocks.append(cur)
v = lambda ins: sum(1 for i in ins if i.startswith('v_'))
print(f'{want}: {sum(v(b[1]) for b in blocks)} static VALU, {sum(1 for b in blocks for i in b[1] if i.startswith("s_"))} SALU, {len(blocks)} blocks')
for name, ins in blocks:
    if v(ins) >= floor:
        mem = [i.split()[0] for i in ins if i.startswith(('buffer_', 'global_', 'ds_', 's_load', 's_buffer', 'flat_', 'scratch_'))]
        div = sum(1 for i in ins if i.startswith(('v_div_', 'v_rcp', 'v_sqrt', 'v_rsq')))
        print(f'{name:>10s}  VALU {v(ins):4d} of {len(ins):4d}  div/sqrt ops {div:3d}  mem: {" ".join(mem[:16])}{" ..." if len(mem) > 16 else ""}')
