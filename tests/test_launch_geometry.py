"""The launch geometry, walked on the CPU: ms_render's plan (ray groups per wave, the one-group waves every XCD's blocks end with)
and the render kernel's own block -> (env, agent, rays) mapping - the same function the device runs - over whole launches: every
ray of every agent of every env cast exactly once, whatever the env count does modulo the eight XCDs; an XCD's envs contiguous,
its wide waves before its one-group ones.  And the packing of envs into physics waves."""
import ctypes as C

import numpy as np
import pytest

from megastep_amd import _lib

SLOTS = 6144                     # MI355X: 256 CUs x 4 SIMDs x 6 waves


def _launch(h, n_envs, n_agents, res, pinned=0, rounds=-1., tail_envs=-1, slots=SLOTS):
    groups = C.c_int(0)
    n_blocks = h.ms_host_render_plan(n_envs, n_agents, res, slots, pinned, rounds, tail_envs, C.byref(groups))
    out = (C.c_int*4)()
    blocks = []
    for b in range(n_blocks):
        ok = h.ms_host_render_block(n_envs, n_agents, res, slots, pinned, rounds, tail_envs, b, out)
        assert ok in (0, 1)
        blocks.append((b, ok, *out) if ok else (b, 0, -1, -1, -1, -1))
    assert h.ms_host_render_block(n_envs, n_agents, res, slots, pinned, rounds, tail_envs, n_blocks, out) == -1
    return groups.value, np.array(blocks, dtype=np.int64).reshape(-1, 6)


def _check(n_envs, n_agents, res, groups, blocks):
    cast = np.zeros((n_envs, n_agents, res), dtype=np.int32)
    live = blocks[blocks[:, 1] == 1]
    for b, _, n, a, r0, span in live:
        assert 0 <= n < n_envs and 0 <= a < n_agents and 0 <= r0 < res and r0 % 64 == 0 and span in (64, 64*groups)
        cast[n, a, r0:min(r0 + span, res)] += 1
    assert (cast == 1).all(), 'every ray of every agent exactly once'
    if groups > 1:
        for x in range(8):                                               # XCD x takes blocks x, x + 8, ...
            mine = live[live[:, 0] % 8 == x]
            if not len(mine):
                continue
            envs = mine[:, 2]
            assert (np.diff(envs) >= 0).all(), 'an XCD walks its envs in order'
            assert set(envs) == set(range(envs.min(), envs.max() + 1)), 'an XCD holds a contiguous run of envs'
            wide = mine[:, 5] == 64*groups
            assert not wide[np.argmax(~wide):].any() if (~wide).any() else True, 'wide waves first, one-group waves behind them'
        # the XCDs' runs of envs tile the envs in order
        firsts = [live[live[:, 0] % 8 == x][:, 2].min() for x in range(8) if (live[:, 0] % 8 == x).any()]
        assert firsts == sorted(firsts)
        spare = blocks[blocks[:, 1] == 0]
        assert len(spare) < 8*n_agents*((res + 63)//64 + 1), 'spare blocks: at most the blocks of one env per XCD'
    else:
        assert len(live) == len(blocks) == n_envs*n_agents*((res + 63)//64)


@pytest.mark.parametrize('n_envs,n_agents,res', [(1, 1, 1), (3, 2, 64), (7, 3, 100), (8, 1, 256), (9, 4, 320), (21, 2, 512), (64, 4, 129), (100, 1, 600), (37, 5, 256)])
def test_every_ray_is_cast_once_under_every_plan(n_envs, n_agents, res):
    h = _lib.lib()
    for pinned in (1, 2, 4):
        for tail_envs in (0, 1, 7, 8, 9, n_envs - 1, n_envs, n_envs + 5, -1):
            if tail_envs < -1:
                continue
            groups, blocks = _launch(h, n_envs, n_agents, res, pinned, -1., tail_envs)
            assert groups == pinned
            _check(n_envs, n_agents, res, groups, blocks)
        for rounds in (0., .001, .5):
            groups, blocks = _launch(h, n_envs, n_agents, res, pinned, rounds, -1, slots=64)
            _check(n_envs, n_agents, res, groups, blocks)


def test_the_rule_for_wide_waves():
    """Four groups per wave from 256 rays up when the launch has two and a half rounds of such waves; one otherwise; never two."""
    h = _lib.lib()
    g = C.c_int(0)

    def plan(n, a, r):
        blocks = h.ms_host_render_plan(n, a, r, SLOTS, 0, -1., -1, C.byref(g))
        return g.value, blocks
    assert plan(4096, 4, 64)[0] == 1 and plan(4096, 4, 128)[0] == 1 and plan(4096, 1, 64) == (1, 4096)
    assert plan(4096, 4, 512)[0] == 4 and plan(32768, 1, 256)[0] == 4 and plan(4096, 4, 256)[0] == 4
    assert plan(4096, 1, 256)[0] == 1, 'two thirds of one round of wide waves: the plain kernel'
    assert plan(2048, 4, 512)[0] == 4 and plan(1024, 4, 512)[0] == 1
    assert plan(4096, 4, 64)[1] == 4096*4 and plan(4096, 4, 128)[1] == 4096*4*2
    # the default tail: half a round of the wide waves' work as one-group waves - 0.5 x 6144 x 4 = 12288 of them
    groups, blocks = _launch(h, 4096, 4, 512)
    single = blocks[(blocks[:, 1] == 1) & (blocks[:, 5] == 64)]
    assert groups == 4 and 12288 <= len(single) < 12288 + 8*4*8
    _check(4096, 4, 512, groups, blocks)


def test_envs_per_physics_wave():
    h = _lib.lib()
    pack = h.ms_host_physics_pack
    assert pack(4096, 4, 1, 0) == 2 and pack(4096, 1, 1, 0) == 2 and pack(3071, 4, 1, 0) == 1 and pack(4096, 4, 0, 0) == 1
    assert pack(32768, 1, 1, 0) == 8 and pack(16384, 4, 1, 0) == 4 and pack(8192, 4, 1, 0) == 2 and pack(262144, 1, 1, 0) == 16
    for n in (1, 100, 4096, 10**5, 10**6):
        for a in (1, 2, 3, 4, 8, 16, 17, 64, 100):
            for gridded in (0, 1):
                k = pack(n, a, gridded, 0)
                assert k >= 1 and (k == 1 or (gridded and k*a <= 32))
                for pinned in (1, 2, 5, 64):
                    kp = pack(n, a, gridded, pinned)
                    assert kp == (pinned if gridded and pinned*a <= 64 else 1)


def test_random_worlds_and_plans():
    """The same walk over randomly drawn worlds, pinnings and tails (hypothesis)."""
    from hypothesis import given, settings, strategies as st
    h = _lib.lib()

    @settings(max_examples=80, deadline=None)
    @given(n_envs=st.integers(1, 200), n_agents=st.integers(1, 6), res=st.integers(1, 700), pinned=st.sampled_from([0, 1, 2, 4]),
           tail_envs=st.integers(-1, 210), rounds=st.sampled_from([-1., 0., .25, .5, 3.]), slots=st.sampled_from([8, 64, 6144]))
    def walk(n_envs, n_agents, res, pinned, tail_envs, rounds, slots):
        groups, blocks = _launch(h, n_envs, n_agents, res, pinned, rounds, tail_envs, slots=slots)
        assert groups in (1, 2, 4) and (pinned == 0 or groups == pinned)
        _check(n_envs, n_agents, res, groups, blocks)
    walk()
