"""bench.py's multi-GPU plumbing without GPUs: two ranks under torch.distributed.run, `--dry-run-cpu` (stub kernels,
real world build, real sharding, real gloo rendezvous, real timing protocol), checked through the JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(nproc, extra=(), self_launch=False, environ=()):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='2', **dict(environ))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    args = ['--gpus', str(nproc), '--steps', '6', '--warmup', '2', '--envs', '24', '--agents', '2', '--res', '16',
            '--unique', '16', '--dry-run-cpu', *extra]
    if nproc == 1 or self_launch:
        cmd = [sys.executable, 'bench.py', *args]
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}',
               '--master-addr', '127.0.0.1', '--master-port', '29613', 'bench.py', *args]
    proc = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, 'exactly one JSON line, from rank 0'
    return json.loads(lines[0])


def test_bench_line_two_ranks_gloo():
    out = _run(2)
    assert out['n_gpus'] == 2 and out['steps'] == 6 and out['warmup'] == 2 and out['scaling'] == 'weak'
    assert out['metric'] == 'env-steps/sec' and out['higher_is_better'] is True and out['vs_baseline'] is None
    cfg = out['config']
    assert cfg['envs_total'] == 48 and 1 <= cfg['envs_this_rank'] < 48          # slices balanced by cost, not count
    # value is the whole job's rate: all envs x steps / the slowest rank's time
    assert abs(out['value'] - 48*6/(out['ms_per_step']*6e-3)) < 1e-6*out['value']
    assert abs(out['agent_steps_per_sec'] - 2*out['value']) < 1e-6*out['value']
    ev = out['eager']['step_ms_hip_events']
    assert ev['min'] <= ev['median'] <= ev['max']
    assert out['roofline']['algorithmic_bytes_per_launch'] > 0 and 'no collectives' in cfg['parallelism']


def test_plain_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` as the driver types it, no torchrun around it: bench.py starts the two ranks itself
    (one process per device, reference common.h:39-41) and rank 0's line says n_gpus 2 and counts both slices."""
    out = _run(2, self_launch=True)
    assert out['n_gpus'] == 2 and out['config']['envs_total'] == 2*24
    pr = out['per_rank']
    assert len(pr['envs']) == 2 and sum(pr['envs']) == 48 and len(pr['ms_per_step']) == 2
    assert all(0 < t <= out['ms_per_step']*1.5 + 1 for t in pr['ms_per_step'])


def test_the_rendezvous_is_not_inside_what_is_timed(tmp_path):
    """`value` is N x K over the slowest rank's OWN synchronize-to-synchronize time. A barrier that takes
    10 ms (two orders of magnitude above a region of stub steps) must leave ms_per_step where the ranks' own times are -
    round 4 read the clock after a closing barrier, which at the driver's --steps 20 would have capped the 8-GPU line."""
    slow = _run(2, environ={'BENCH_TEST_BARRIER_SLEEP_MS': '10'})
    own = slow['per_rank']['ms_per_step']
    assert len(own) == 2 and 'OWN' in slow['timing']
    # 10 ms of barrier over 6 steps would be 1.7 ms per step; a stub step is microseconds
    assert slow['ms_per_step'] < 0.5, slow['ms_per_step']
    # the median region's MAX over ranks against the ranks' median regions: the same quantity up to the noise of a
    # microsecond-long CPU region (median of maxima vs maximum of medians)
    assert slow['ms_per_step'] <= 1.05*max(own) + 0.02, (slow['ms_per_step'], own)
    assert slow['timed_regions']['ms_per_step_min'] <= slow['ms_per_step'] <= slow['timed_regions']['ms_per_step_max']
    assert [h['rank'] for h in slow['per_rank']['host']] == [0, 1] and slow['per_rank']['host'][0]['omp_num_threads'] == '2'
    # scaling efficiency against an N = 1 line of the same command, when one is handed over
    line = tmp_path/'n1.json'
    one = _run(1)
    line.write_text(json.dumps(one) + '\n')
    two = _run(2, ('--baseline-line', str(line)))
    eff = two['scaling_efficiency']
    assert eff['n1_value'] == one['value'] and abs(eff['efficiency'] - two['value']/(2*one['value'])) < 1e-9


def test_world_size_must_match_the_gpus_asked_for():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', WORLD_SIZE='1', RANK='0')
    proc = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--dry-run-cpu', '--envs', '8', '--unique', '8'], cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=300)
    assert proc.returncode != 0 and 'WORLD_SIZE' in proc.stderr


def test_bench_line_single_rank():
    out = _run(1, ('--no-graph',))
    assert out['n_gpus'] == 1 and out['config']['envs_total'] == 24 and out['config']['envs_this_rank'] == 24
    assert out['per_rank']['envs'] == [24] and len(out['per_rank']['ms_per_step']) == 1
    assert 'eager' in out['config']['launch']


def test_every_rank_builds_its_own_slice_and_the_slices_make_the_world(monkeypatch):
    """bench.build_world with world > 1 (reference: common.h:136-144 slices, it does not replicate): what reaches a rank's
    device is its slice and nothing else, the cuts are the cost-balanced ones every rank works out alike from the
    floorplans, and the slices laid end to end are the single-rank build of the same job bit for bit."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from megastep_amd import scene, sharding
    built = []
    real = scene.scenery

    def spy(geometries, *args, **kwargs):
        result = real(geometries, *args, **kwargs)
        built.append((len(geometries), kwargs.get('envs'), len(result.lines), result.lines.vals.shape[0], result.textures.vals.shape[0]))
        return result
    monkeypatch.setattr(scene, 'scenery', spy)
    kw = dict(n_agents=2, res=16, fov=130., device=torch.device('cpu'), seed=1, n_unique=16, bake=False)
    world = 3
    whole, geometries = bench.build_world(3*24, world=1, **kw)            # the same 72-env job on one rank
    parts = [bench.build_world(24, rank=r, world=world, **kw) for r in range(world)]
    cuts = [bench.rank_slice(geometries, 2, 16, r, world) for r in range(world)]
    assert cuts[0][0] == 0 and cuts[-1][1] == 72 and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    # balanced by lines x agents x rays, not by env count: each rank's share of the cost within one env of a third
    cost = np.array([2*8 + len(g['walls']) for g in geometries], float)
    for (a, b) in cuts:
        assert abs(cost[a:b].sum() - cost.sum()/world) <= cost.max()
    for (n_geoms, envs, n_envs, n_lines, n_texels), (core, geoms), (a, b) in zip(built[1:], parts, cuts):
        # the build was handed the whole job's floorplans but made tensors for its slice only
        assert n_geoms == 72 and tuple(envs) == (a, b) and n_envs == b - a == core.n_envs == len(geoms)
        assert n_lines == int(whole.scenery.lines.widths[a:b].sum()) < whole.scenery.lines.vals.shape[0]
        assert n_texels < whole.scenery.textures.vals.shape[0]
    for k in ('lines', 'lights', 'textures'):
        for f in ('vals', 'widths'):
            assert torch.equal(torch.cat([getattr(getattr(c.scenery, k), f) for c, _ in parts]), getattr(getattr(whole.scenery, k), f)), (k, f)
    # the same cuts as sharding a built world by its measured cost would make
    for r in range(world):
        assert sharding.env_slice(72, r, world, sharding.render_cost(whole.scenery, 16)) == cuts[r]
