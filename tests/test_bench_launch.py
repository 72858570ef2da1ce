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
