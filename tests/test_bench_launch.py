"""bench.py's rank handling, the parts that need no GPU: `--gpus N` without a launcher starts its own ranks (and says so when the
node has fewer devices), and a mismatch between `--gpus` and the launcher's WORLD_SIZE is refused instead of mislabelled."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=('WORLD_SIZE', 'RANK', 'LOCAL_RANK')):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, cwd=ROOT, env=env,
                          timeout=300)


def test_gpus_flag_without_enough_devices_is_a_clear_error():
    import torch
    have = torch.cuda.device_count()          # does not initialise the GPU
    r = _run(['--gpus', str(have + 2), '--steps', '1', '--warmup', '0'])
    assert r.returncode != 0
    assert 'needs %d GPUs on this node, found %d' % (have + 2, have) in (r.stderr + r.stdout)


def test_gpus_flag_must_match_world_size():
    r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0'], env_extra={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'}, drop=())
    assert r.returncode != 0
    assert '--gpus 2 but WORLD_SIZE=1' in (r.stderr + r.stdout)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_test_hooks_in_the_environment_are_refused_without_the_flag():
    """MVF_BENCH_BACKEND / MVF_BENCH_SHARE_GPU swap the interconnect / stack the ranks on one GPU: a measurement run must not pick
    them up from a stray environment."""
    for k, v in (('MVF_BENCH_BACKEND', 'gloo'), ('MVF_BENCH_SHARE_GPU', '1')):
        r = _run(['--steps', '1', '--warmup', '0'], env_extra={k: v})
        assert r.returncode != 0
        assert k in (r.stderr + r.stdout) and '--test-hooks' in (r.stderr + r.stdout)


def test_two_ranks_gloo_plumbing_reports_the_group_as_it_ran():
    """The N > 1 control flow of bench.py on the CPU (no kernel, `value` null): gloo group of two ranks, the benchmark's gradient
    buckets all-reduced once through GradReducer; rank 0's line carries the backend and world size THE PROCESS GROUP reports and
    the all-reduce payload of a step (the trainable parameters' flat fp32 gradient buffer, 4.8 M elements)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['MVF_BENCH_BACKEND'] = 'gloo'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--test-hooks', '--plumbing']
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    c = j['config']
    assert j['n_gpus'] == 2 and j['value'] is None and j['plumbing'] is True
    assert c['backend'] == 'gloo' and c['world_size_seen'] == 2 and c['max_rank_seen'] == 1 and c['parallelism'] == 'dp2'
    assert 18e6 < c['comm']['allreduce_bytes_per_step'] < 21e6 and c['comm']['buckets'] >= 1


def test_a_rank_whose_peer_never_arrives_exits_with_its_stage_instead_of_hanging():
    """--init-timeout: rank 0 of a two-rank group whose rank 1 is never started must give up with exit code 4 and say where."""
    r = _run(['--gpus', '2', '--test-hooks', '--plumbing', '--init-timeout', '6'],
             env_extra={'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(_free_port()),
                        'MVF_BENCH_BACKEND': 'gloo'}, drop=())
    assert r.returncode == 4, (r.returncode, r.stderr[-1500:])
    assert 'last stage: init_process_group(gloo)' in r.stderr
