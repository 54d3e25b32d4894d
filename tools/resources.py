"""Registers, scratch, occupancy and LDS of every kernel of libmegastep_hip.so, as hipcc's -Rpass-analysis=kernel-resource-usage
reports them for gfx950 (no GPU needed):

    python tools/resources.py [extra hipcc flags, e.g. -DMS_VCAP_WIDE=96]  >  profiles/rNN_resources.txt

Uses the product's own flags (megastep_amd/csrc/Makefile's FLAGS), so what it prints is what `make` builds.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'megastep_amd', 'csrc')


def flags():
    text = open(os.path.join(CSRC, 'Makefile')).read().replace('\\\n', ' ')
    line = re.search(r'^FLAGS = (.*)$', text, re.M).group(1)
    return line.replace('$(ARCH)', 'gfx950').split()


def main(extra):
    cmd = ['/opt/rocm/bin/hipcc', *flags(), *extra, '-Rpass-analysis=kernel-resource-usage', '-o', '/tmp/ms_resources.so', 'megastep_hip.hip']
    out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if out.returncode:
        sys.exit(out.stderr[-4000:])
    rows, cur = [], None
    for l in out.stderr.splitlines():
        m = re.search(r'remark:\s+(?:Function )?Name: (\S+)', l)
        if m:
            cur = {'name': m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r'remark:\s+(VGPRs|AGPRs|TotalSGPRs|ScratchSize|Occupancy|LDS Size)(?: \[[^\]]*\])?: (\d+)', l)
        if m and cur is not None:
            cur[m.group(1)] = m.group(2)
    names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.split('\n')
    print(f'# hipcc {" ".join(extra)}' if extra else '# product flags')
    print(f'{"kernel":44s} {"VGPRs":>5s} {"SGPRs":>5s} {"scratch B/lane":>14s} {"waves/SIMD":>10s} {"LDS B/block":>11s}')
    for r, n in zip(rows, names):
        n = n.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        print(f'{n:44s} {r.get("VGPRs", "?"):>5s} {r.get("TotalSGPRs", "?"):>5s} {r.get("ScratchSize", "?"):>14s} {r.get("Occupancy", "?"):>10s} {r.get("LDS Size", "?"):>11s}')


if __name__ == '__main__':
    main(sys.argv[1:])
