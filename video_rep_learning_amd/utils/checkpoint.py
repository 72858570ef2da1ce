"""Checkpoint files in the reference's on-disk layout (what CARL_MVF/models/__init__.py:17-60 reads and writes):

    <LOGDIR>/checkpoints/checkpoint_epoch_<5-digit epoch>.pth  =  torch.save({
        'epoch': int, 'model_state': <unwrapped model>.state_dict(), 'optimizer_state': optimizer.state_dict(), 'cfg': ...})

so that a run started with the reference resumes here and vice versa (tests/test_checkpoint.py).  Differences in
behaviour, none in format: files are written to a temporary name and renamed (a killed job never leaves a truncated
"latest" checkpoint), and the stored cfg is a plain dict (no argparse namespace inside the pickle)."""
import os
import re
import tempfile

import torch

from . import logging

logger = logging.get_logger(__name__)

_PATTERN = re.compile(r'checkpoint_epoch_(\d+)\.pth$')


def directory(cfg):
    return os.path.join(cfg.LOGDIR, 'checkpoints')


def epoch_file(cfg, epoch):
    return os.path.join(directory(cfg), 'checkpoint_epoch_%05d.pth' % epoch)


def bare(model):
    """The module whose state dict is stored: DDP-style wrappers keep it in `.module`."""
    return getattr(model, 'module', model)


def latest(cfg):
    """Path of the newest checkpoint of this LOGDIR or None.  The reference sorts the file NAMES (zero-padded epochs make
    that the numeric order); files that do not match the naming scheme but contain 'checkpoint' sort as it would, too."""
    d = directory(cfg)
    if not os.path.isdir(d):
        return None
    names = sorted(n for n in os.listdir(d) if 'checkpoint' in n and not n.endswith('.tmp'))
    return os.path.join(d, names[-1]) if names else None


def write(cfg, model, optimizer, epoch):
    os.makedirs(directory(cfg), exist_ok=True)
    payload = {'epoch': epoch, 'model_state': bare(model).state_dict(), 'optimizer_state': optimizer.state_dict(),
               'cfg': {k: v for k, v in cfg.items() if k != 'args'}}
    target = epoch_file(cfg, epoch)
    fd, tmp = tempfile.mkstemp(dir=directory(cfg), suffix='.tmp')
    os.close(fd)
    torch.save(payload, tmp)
    os.replace(tmp, target)
    logger.info('Saving epoch %d checkpoint at %s', epoch, target)
    return target


def _read(path):
    return torch.load(path, map_location='cpu', weights_only=False)


def restore(cfg, model, optimizer):
    """Resume from this LOGDIR's newest checkpoint (-> next epoch), else initialise the weights from
    cfg.MODEL.PRETRAINED_CHECKPOINT if one is named (-> epoch 0), else leave everything as constructed (-> 0)."""
    path = latest(cfg)
    if path is not None:
        logger.info('Loading checkpoint at %s', path)
        state = _read(path)
        bare(model).load_state_dict(state['model_state'])
        optimizer.load_state_dict(state['optimizer_state'])
        return state['epoch'] + 1
    warm = cfg.MODEL.get('PRETRAINED_CHECKPOINT', None) if hasattr(cfg.MODEL, 'get') else None
    if warm is not None:
        if not os.path.exists(warm):
            print('ERROR: invalid path specified for cfg.MODEL.PRETRAINED_CHECKPOINT')
            print('could not find checkpoint at: ' + warm)
            raise SystemExit(-1)
        logger.info('Loading pretrained checkpoint at %s', warm)
        bare(model).load_state_dict(_read(warm)['model_state'])
    return 0
