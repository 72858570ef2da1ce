"""BASELINE configs[0] ("configs_mvf/penn_mvf.yml, 8 frames, batch 1, world_size=1 on CPU/gloo -- plumbing, runs without a GPU"):
`train.main --plumbing --device cpu --backend gloo` runs everything AROUND the kernels (reference: CARL_MVF/train.py:230-307 --
parser, config merge, process group, build_model, construct_optimizer, loader, checkpoint save / restore, the iteration's
collectives) and must stop with MvfError at the first HIP call: the product has no CPU compute path."""
import os
import socket

import pytest
import torch
import yaml

from video_rep_learning_amd import train
from video_rep_learning_amd.utils import presets
from video_rep_learning_amd.utils.parser import to_dict


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_config0_plumbing_on_cpu_gloo(tmp_path, monkeypatch):
    cfg_file = str(tmp_path / 'penn_mvf.yml')                      # configs_mvf/penn_mvf.yml as shipped (tests/test_config.py)
    with open(cfg_file, 'w') as f:
        yaml.safe_dump(to_dict(presets.penn_mvf()), f)
    for k, v in dict(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), WORLD_SIZE='1', RANK='0', LOCAL_RANK='0').items():
        monkeypatch.setenv(k, v)
    argv = ['--cfg_file', cfg_file, '--logdir', str(tmp_path / 'log'), '--synthetic', '--device', 'cpu', '--backend', 'gloo',
            '--plumbing', '--opts', 'TRAIN.NUM_FRAMES', '8', 'TRAIN.BATCH_SIZE', '1', 'TRAIN.MAX_EPOCHS', '1']
    out = train.main(argv)
    assert out['plumbing'] is True
    assert 'cpu' in out['stopped_at'] and 'no CPU fallback' in out['stopped_at'] or 'gfx950' in out['stopped_at'], out
    assert not torch.distributed.is_initialized()
    # what the run left behind: the stored config -- and NO checkpoint: the round trip (reference naming and layout) happens in
    # a scratch directory that is removed, so that a LOGDIR holding checkpoint E never gains an "E + 1" with epoch-E weights
    assert os.path.exists(tmp_path / 'log' / 'config.yml')
    ck = out['checkpoint']
    assert ck['file'] == 'checkpoint_epoch_00000.pth' and set(ck['keys']) == {'epoch', 'model_state', 'optimizer_state', 'cfg'}
    assert ck['epoch'] == 0 and {'embed', 'backbone'} <= set(ck['model_prefixes'])
    assert not os.path.exists(tmp_path / 'log' / 'plumbing_scratch') and not os.path.exists(tmp_path / 'log' / 'checkpoints')


def test_plumbing_flag_is_not_a_cpu_training_mode(tmp_path, monkeypatch):
    """Without --plumbing the same command must fail at the first kernel call instead of training on the CPU."""
    from video_rep_learning_amd._lib import MvfError
    cfg_file = str(tmp_path / 'penn_mvf.yml')
    with open(cfg_file, 'w') as f:
        yaml.safe_dump(to_dict(presets.penn_mvf()), f)
    for k, v in dict(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), WORLD_SIZE='1', RANK='0', LOCAL_RANK='0').items():
        monkeypatch.setenv(k, v)
    argv = ['--cfg_file', cfg_file, '--logdir', str(tmp_path / 'log'), '--synthetic', '--device', 'cpu', '--backend', 'gloo',
            '--max_iters', '1', '--opts', 'TRAIN.NUM_FRAMES', '8', 'TRAIN.BATCH_SIZE', '1', 'TRAIN.MAX_EPOCHS', '1']
    try:
        with pytest.raises(MvfError):
            train.main(argv)
    finally:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
