"""The C-ABI library: loads, exports every symbol include/megastep_hip.h declares, validates arguments, and fails loudly.
No kernel is launched here (no GPU in the authoring container)."""
import ctypes as C
C_int = C.c_int
import os
import re
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(headers=('megastep_hip.h', 'megastep_hip_test.h')):
    names = set()
    for h in headers:
        text = open(os.path.join(ROOT, 'include', h)).read()
        text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
        names |= set(re.findall(r'\b(ms_[a-z_]+)\s*\(', text))
    return sorted(names)


def test_the_boundary_header_holds_no_test_hooks():
    """include/megastep_hip.h is what a maintainer binds: the entry points that replace wrappers.cpp's, nothing else."""
    assert not [n for n in declared_symbols(('megastep_hip.h',)) if n.startswith(('ms_host_', 'ms_debug_'))]
    assert {'ms_bake', 'ms_physics', 'ms_render'} <= set(declared_symbols(('megastep_hip.h',)))


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
    src = '#include <stdio.h>\n#include "megastep_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu", sizeof(MsConfig), sizeof(MsScenery), sizeof(MsAgents), sizeof(MsRender), sizeof(MsMovement), sizeof(MsStepExtras), sizeof(MsDeathmatch), sizeof(MsExplorer));}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 't.c'), 'w').write(src)
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), os.path.join(d, 't.c'), '-o', os.path.join(d, 't')])
        sizes = list(map(int, subprocess.check_output([os.path.join(d, 't')]).split()))
    assert sizes == [C.sizeof(_lib.MsConfig), C.sizeof(_lib.MsScenery), C.sizeof(_lib.MsAgents), C.sizeof(_lib.MsRender),
                     C.sizeof(_lib.MsMovement), C.sizeof(_lib.MsStepExtras), C.sizeof(_lib.MsDeathmatch), C.sizeof(_lib.MsExplorer)]


def test_bad_arguments_are_rejected_before_any_launch():
    from megastep_amd import _lib
    h = _lib.lib()
    cfg = _lib.MsConfig(.1, 64, 130., 10.)
    assert h.ms_physics(None, None, None, C.byref(cfg), None) == -1
    assert h.ms_render(None, None, None, C.byref(cfg), None) == -1
    assert h.ms_bake(None, None, None) == -1
    assert h.ms_deathmatch_shoot(4, 4, None, None) == -1 and h.ms_deathmatch_shoot(0, 4, C.byref(_lib.MsDeathmatch()), None) == -1
    assert h.ms_deathmatch_shoot(4, 4, C.byref(_lib.MsDeathmatch()), None) == -1          # (null tensors)
    assert h.ms_explorer_books(4, None, None) == -1 and h.ms_explorer_books(4, C.byref(_lib.MsExplorer()), None) == -1
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
    health = torch.ones((1, 1))
    books = [torch.zeros(1, dtype=torch.int32) for _ in range(4)]
    for call in (lambda: cuda.bake(scenery), lambda: cuda.physics(scenery, agents), lambda: cuda.render(scenery, agents),
                 lambda: cuda.step_render(scenery, agents),
                 lambda: cuda.deathmatch_shoot(torch.zeros((1, 1, 2), dtype=torch.int32), torch.zeros((1, 1, 2)), torch.zeros((1, 2)), health,
                                               health.clone(), torch.zeros((1, 1), dtype=torch.bool)),
                 lambda: cuda.explorer_books(*books, torch.zeros(1, dtype=torch.bool), 200, 64)):
        with pytest.raises(RuntimeError, match='GPU'):
            call()
    with pytest.raises(RuntimeError, match='N, A'):
        cuda.deathmatch_shoot(torch.zeros((1, 1, 2), dtype=torch.int32), torch.zeros((1, 1, 2)), torch.zeros((2, 2)), health, health.clone(),
                              torch.zeros((1, 1), dtype=torch.bool))
    with pytest.raises(RuntimeError, match=r'\(N,\)'):
        cuda.explorer_books(*books, torch.zeros(2, dtype=torch.bool), 200, 64)


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under megastep_amd/ may import, load, link or execute it."""
    banned = [r'^\s*(from|import)\s+oracle\b', r'libmegastep_oracle', r'oracle/', r'oracle_[a-z]+\s*\(', r'#include\s*"[^"]*oracle']
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'megastep_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp', 'Makefile')):
                text = open(os.path.join(dirpath, f)).read()
                for pat in banned:
                    assert not re.search(pat, text, flags=re.M), (os.path.join(dirpath, f), pat)


def test_bake_bins_never_hide_an_obstructing_wall():
    """ms_bake's angular bins (host instantiations of the device functions): whenever the reference's obstructed()
    test (kernels.cu:238-259, evaluated in binary32) says a wall blocks a point from a light, the point's bin lies
    in the wall's run of bins - over random scenes salted with the degenerate cases (points and walls next to the
    light, lights on a wall's line, tiny and huge walls)."""
    from megastep_amd import _lib
    h = _lib.lib()
    rng = np.random.RandomState(0)
    n = 60000
    f = np.float32
    I = rng.uniform(1, 20, (n, 2)).astype(f)
    a = rng.uniform(0, 21, (n, 2)).astype(f)
    v = (rng.normal(size=(n, 2))*rng.choice([.01, .3, 3., 15.], (n, 1))).astype(f)
    pt = rng.uniform(0, 21, (n, 2)).astype(f)
    k = n//6
    pt[:k] = I[:k] + (rng.normal(size=(k, 2))*rng.choice([1e-4, 1e-2, .1], (k, 1))).astype(f)         # points by the light
    a[k:2*k] = I[k:2*k] + (rng.normal(size=(k, 2))*rng.choice([1e-4, 1e-2, .1], (k, 1))).astype(f)  # walls by the light
    t = rng.uniform(-.2, 1.2, (k, 1)).astype(f)                                                      # lights on the wall's line
    I[2*k:3*k] = (a[2*k:3*k] + t*v[2*k:3*k] + (rng.normal(size=(k, 2))*rng.choice([0, 1e-5, 1e-3], (k, 1)))).astype(f)
    pt[3*k:4*k] = (a[3*k:4*k] + rng.uniform(0, 1, (k, 1)).astype(f)*v[3*k:4*k]                         # points just behind walls
                  + (rng.normal(size=(k, 2))*1e-2)).astype(f)
    b = (a + v).astype(f)
    V = (b - a).astype(f)
    # obstructed(): intersect(I, C - I, a, b - a), binary32 as the reference writes it
    U = (pt - I).astype(f)
    cross = lambda p, q: (p[:, 0]*q[:, 1] - p[:, 1]*q[:, 0]).astype(f)
    UxV = cross(U, V)
    PQ = (a - I).astype(f)
    with np.errstate(all='ignore'):
        s = (cross(PQ, V)/UxV).astype(f)
        tt = (cross(PQ, U)/UxV).astype(f)
    blocked = (np.abs(UxV) >= f(1e-3)) & (tt > 0) & (tt < 1) & (s > 0) & (s < f(.999))
    assert blocked.sum() > 2000
    first, count = C_int(), C_int()
    narrow = 0
    for i in np.flatnonzero(blocked):
        pb = h.ms_host_bake_point_bin(I[i, 0], I[i, 1], pt[i, 0], pt[i, 1])
        h.ms_host_bake_wall_bins(I[i, 0], I[i, 1], a[i, 0], a[i, 1], b[i, 0], b[i, 1], C.byref(first), C.byref(count))
        assert 0 <= first.value < 64 and 1 <= count.value <= 64
        narrow += count.value < 64
        assert pb == -1 or (pb - first.value) % 64 < count.value, (i, pb, first.value, count.value)
    assert narrow > 1000        # the bins do cull


def test_a_compile_error_is_never_papered_over_with_the_library_on_disk(monkeypatch):
    """Sources that differ from what the in-tree library was built from and do not compile: lib() must raise, not load
    yesterday's kernels (a missing toolchain is the one case in which the library on disk is taken as it is)."""
    import subprocess
    import types
    from megastep_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, '_source_hash', lambda: 'edited')
    monkeypatch.delenv('MEGASTEP_HIP_LIB', raising=False)
    monkeypatch.setattr(subprocess, 'run', lambda *a, **k: types.SimpleNamespace(returncode=1, stdout='', stderr='error: expected ;'))
    with pytest.raises(RuntimeError, match='hipcc build of libmegastep_hip.so failed'):
        _lib.lib()
    # ... whereas without the tools the library that is there is loaded, with a warning
    monkeypatch.setenv('HIPCC', '/nonexistent/hipcc')
    with pytest.warns(UserWarning, match='could not be rebuilt'):
        assert _lib.lib().ms_abi_version() == _lib.ABI_VERSION
