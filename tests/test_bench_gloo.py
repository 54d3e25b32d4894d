"""bench.py's multi-GPU plumbing without GPUs: two ranks under torch.distributed.run, `--dry-run-cpu` (stub kernels,
real world build, real sharding, real gloo rendezvous, real timing protocol), checked through the JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(nproc, extra=()):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='2')
    args = ['--gpus', str(nproc), '--steps', '6', '--warmup', '2', '--envs', '24', '--agents', '2', '--res', '16',
            '--unique', '16', '--dry-run-cpu', *extra]
    if nproc == 1:
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


def test_bench_line_single_rank():
    out = _run(1, ('--no-graph',))
    assert out['n_gpus'] == 1 and out['config']['envs_total'] == 24 and out['config']['envs_this_rank'] == 24
    assert 'eager' in out['config']['launch']
