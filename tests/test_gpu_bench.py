"""bench.py as the driver launches it: single process, and two ranks through torch.distributed.run (here both on the one
GPU of the test box with the gloo backend -- RCCL refuses two ranks on one device; the control flow, the collectives'
ORDER on every rank and the output contract are what is tested, not the interconnect)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
        'dtype', 'data', 'config', 'roofline'}


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_process_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '2', '--profile-steps', '1',
                        '--no-cpu-baseline'], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j['config']['backend'] is None and j['config']['world_size_seen'] == 1 and 'comm' not in j['config']
    assert KEYS <= set(j) and j['n_gpus'] == 1 and j['steps'] == 3 and j['warmup'] == 2 and j['vs_baseline'] is None
    assert j['higher_is_better'] is True and j['scaling'] == 'weak' and j['unit'] == 'clips/s' and j['value'] > 0
    rf = j['roofline']
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(rf) and 0 < rf['frac'] < 1
    assert 'workload' in j['config'] and 'model' not in j['config']


def test_bench_two_ranks_do_not_deadlock():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MVF_BENCH_SHARE_GPU='1', MVF_BENCH_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2',
           '--profile-steps', '1', '--test-hooks']
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    j = _line(r.stdout)
    c = j['config']
    assert j['n_gpus'] == 2 and c['parallelism'] == 'dp2' and j['value'] > 0
    # the line says what the process group WAS, and what the gradient all-reduce moved / cost the compute stream
    assert c['backend'] == 'gloo' and c['world_size_seen'] == 2
    assert 18e6 < c['comm']['allreduce_bytes_per_step'] < 21e6 and c['comm']['exposed_allreduce_ms_per_step'] >= 0.0


def test_bench_forced_one_rank_rccl_group_reserves_cus_for_the_collectives():
    """MVF_FORCE_REDUCER=1: a ONE-rank `nccl` (= RCCL) group with every collective of the data-parallel step issued, as the
    driver's N > 1 runs issue them; the persistent GEMM must then run under the CU budget that leaves 8 CUs to RCCL."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MVF_FORCE_REDUCER='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '2', '--profile-steps', '1',
                        '--no-cpu-baseline'], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert j['n_gpus'] == 1 and j['config']['gemm_cu_budget'] == cus - 8 and j['value'] > 0
    assert j['config']['backend'] == 'nccl' and j['config']['world_size_seen'] == 1
    assert j['config']['comm']['allreduce_bytes_per_step'] > 18e6


def _device_count():
    import torch
    return torch.cuda.device_count()      # does not initialise the GPU in this process


@pytest.mark.skipif(_device_count() < 2, reason='needs two GPUs: `python bench.py --gpus 2` over RCCL, one rank per GPU')
def test_bench_gpus_2_bare_over_rccl():
    """`python bench.py --gpus 2` exactly as a user (or the driver, through torch.distributed.run) starts it on a multi-GPU
    node: the script launches its own two ranks, backend nccl (= RCCL over xGMI)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2',
                        '--profile-steps', '1'], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    j = _line(r.stdout)
    assert j['n_gpus'] == 2 and j['config']['parallelism'] == 'dp2' and j['value'] > 0
    assert j['config']['backend'] == 'nccl' and j['config']['world_size_seen'] == 2
    assert j['config']['comm']['exposed_allreduce_ms_per_step'] is not None
    assert isinstance(j['config']['gemm_cu_budget'], int)
