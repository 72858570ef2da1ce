"""train.main end to end on the device with the reference's CLI: two epochs of a tiny config on RAW synthetic frames (the
GPU-side augmentation runs inside the loop, datasets/augment.py), validation, checkpoint, and --continue_train resume."""
import glob
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_main_raw_frames_checkpoint_resume(tmp_path):
    import yaml
    from video_rep_learning_amd import train
    from video_rep_learning_amd.utils import presets
    from video_rep_learning_amd.utils.parser import to_dict
    cfg_file = str(tmp_path / 'penn_mvf.yml')                      # the shipped configs_mvf/penn_mvf.yml, as a preset
    with open(cfg_file, 'w') as f:
        yaml.safe_dump(to_dict(presets.penn_mvf()), f)
    logdir = str(tmp_path / 'run')
    argv = ['--cfg_file', cfg_file, '--logdir', logdir, '--synthetic', '--synthetic_raw', '40', '52', '--max_iters', '3', '--opts',
            'MODEL.BASE_MODEL.NETWORK', 'TIMM-vit_small_patch16_224.dino', 'TRAIN.NUM_FRAMES', '8', 'TRAIN.BATCH_SIZE', '2',
            'IMAGE_SIZE', '32', 'TRAIN.MAX_EPOCHS', '2', 'EVAL.VAL_INTERVAL', '1', 'MI355X.COMPUTE_DTYPE', 'fp32']
    train.main(argv)
    cks = sorted(glob.glob(os.path.join(logdir, 'checkpoints', '*.pth')))
    assert [os.path.basename(c) for c in cks] == ['checkpoint_epoch_00001.pth']
    ck = torch.load(cks[0], map_location='cpu', weights_only=False)
    assert ck['epoch'] == 1 and len(ck['optimizer_state']['state']) > 0
    assert int(float(ck['optimizer_state']['state'][0]['step'])) == 6          # 2 epochs x 3 iterations
    rows = [json.loads(l) for l in open(os.path.join(logdir, 'train_logs', 'scalars.jsonl'))]
    losses = [r['value'] for r in rows if r['tag'].startswith('train/loss') or r['tag'] == 'train/loss']
    assert losses and all(v == v and abs(v) < 1e3 for v in losses)            # finite
    # resume: one more epoch starts from the stored state (epoch counter, Adam step count)
    argv2 = [a if a != '2' or argv[i - 1] != 'TRAIN.MAX_EPOCHS' else '3' for i, a in enumerate(argv)] + []
    train.main(['--continue_train', '--tempcfg'] + argv2)
    cks = sorted(glob.glob(os.path.join(logdir, 'checkpoints', '*.pth')))
    assert os.path.basename(cks[-1]) == 'checkpoint_epoch_00002.pth'
    ck2 = torch.load(cks[-1], map_location='cpu', weights_only=False)
    assert int(float(ck2['optimizer_state']['state'][0]['step'])) == 9
