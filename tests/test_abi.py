"""The C-ABI library: loads, exports every symbol include/megastep_hip.h declares, validates arguments, and fails loudly.
No kernel is launched here (no GPU in the authoring container)."""
import ctypes as C
import os
import re
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'megastep_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ms_[a-z_]+)\s*\(', text)))


def test_header_and_loader_agree():
    from megastep_amd import _lib
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    from megastep_amd import _lib
    handle = _lib.lib()
    for name in declared_symbols():
        assert hasattr(handle, name), name
    assert handle.ms_abi_version() == _lib.ABI_VERSION
    assert handle.ms_strerror(0) == b'ok'
    assert b'invalid' in handle.ms_strerror(-1)


def test_struct_layouts_match_the_header():
    """ctypes mirrors of the structs must have the C layout: check sizes against a C compile of the header."""
    import subprocess, tempfile
    from megastep_amd import _lib
    src = '#include <stdio.h>\n#include "megastep_hip.h"\nint main(){printf("%zu %zu %zu %zu", sizeof(MsConfig), sizeof(MsScenery), sizeof(MsAgents), sizeof(MsRender));}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 't.c'), 'w').write(src)
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), os.path.join(d, 't.c'), '-o', os.path.join(d, 't')])
        sizes = list(map(int, subprocess.check_output([os.path.join(d, 't')]).split()))
    assert sizes == [C.sizeof(_lib.MsConfig), C.sizeof(_lib.MsScenery), C.sizeof(_lib.MsAgents), C.sizeof(_lib.MsRender)]


def test_bad_arguments_are_rejected_before_any_launch():
    from megastep_amd import _lib
    h = _lib.lib()
    cfg = _lib.MsConfig(.1, 64, 130., 10.)
    assert h.ms_physics(None, None, None, C.byref(cfg), None) == -1
    assert h.ms_render(None, None, None, C.byref(cfg), None) == -1
    assert h.ms_bake(None, None, None) == -1
    sc, ag, out = _lib.MsScenery(), _lib.MsAgents(), _lib.MsRender()
    assert h.ms_physics(C.byref(sc), C.byref(ag), None, C.byref(cfg), None) == -1
    with pytest.raises(RuntimeError, match='invalid argument'):
        _lib.check(h.ms_render(C.byref(sc), C.byref(ag), C.byref(out), C.byref(cfg), None))


def test_kernel_sincospi_is_bitwise_the_oracles(oracle):
    """The product's sin/cos(pi x) (host instantiation of the device function) against the oracle's restatement."""
    from megastep_amd import _lib
    h = _lib.lib()
    xs = np.concatenate([np.linspace(-3, 3, 4001), np.random.RandomState(1).uniform(-720, 720, 3000)/180]).astype(np.float32)
    for x in xs:
        s, c = C.c_float(), C.c_float()
        h.ms_host_sincospi(float(x), C.byref(s), C.byref(c))
        assert (np.float32(s.value), np.float32(c.value)) == oracle.sincospi(x), x


def test_compute_entry_points_refuse_cpu_tensors():
    import torch
    from megastep_amd import cuda, core, scene, toys
    scenery = scene.scenery([toys.box()], 1, device='cpu', bake=False)
    cuda.initialize(core.AGENT_RADIUS, 64, 130, 10)
    agents = core._init_agents(1, 1, 'cpu')
    for call in (lambda: cuda.bake(scenery), lambda: cuda.physics(scenery, agents), lambda: cuda.render(scenery, agents)):
        with pytest.raises(RuntimeError, match='GPU'):
            call()


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under megastep_amd/ may import, load, link or execute it."""
    banned = [r'^\s*(from|import)\s+oracle\b', r'libmegastep_oracle', r'oracle/', r'oracle_[a-z]+\s*\(', r'#include\s*"[^"]*oracle']
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'megastep_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp', 'Makefile')):
                text = open(os.path.join(dirpath, f)).read()
                for pat in banned:
                    assert not re.search(pat, text, flags=re.M), (os.path.join(dirpath, f), pat)
