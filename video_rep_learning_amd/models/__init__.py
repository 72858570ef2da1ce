"""Model registry + checkpoint I/O with the reference's surface (CARL_MVF/models/__init__.py:8-60)."""
import os

import torch

from ..utils import logging
from .transformer import TransformerModel

logger = logging.get_logger(__name__)


def build_model(cfg, local_rank=None):
    if cfg.MODEL.EMBEDDER_TYPE == 'transformer':
        return TransformerModel(cfg, local_rank)
    raise NotImplementedError("MODEL.EMBEDDER_TYPE '%s': only 'transformer' (MV-Former) is on the MI355X path"
                              % cfg.MODEL.EMBEDDER_TYPE)


def _unwrap(model):
    return model.module if hasattr(model, 'module') else model


def save_checkpoint(cfg, model, optimizer, epoch):
    path = os.path.join(cfg.LOGDIR, 'checkpoints')
    os.makedirs(path, exist_ok=True)
    ckpt_path = os.path.join(path, 'checkpoint_epoch_{:05d}.pth'.format(epoch))
    checkpoint = {
        'epoch': epoch,
        'model_state': _unwrap(model).state_dict(),
        'optimizer_state': optimizer.state_dict(),
        'cfg': {k: v for k, v in cfg.items() if k != 'args'},
    }
    torch.save(checkpoint, ckpt_path)
    logger.info(f'Saving epoch {epoch} checkpoint at {ckpt_path}')


def load_checkpoint(cfg, model, optimizer):
    path = os.path.join(cfg.LOGDIR, 'checkpoints')
    if os.path.exists(path):
        names = [f for f in os.listdir(path) if 'checkpoint' in f]
        if len(names) > 0:
            ckpt_path = os.path.join(path, sorted(names)[-1])
            logger.info(f'Loading checkpoint at {ckpt_path}')
            checkpoint = torch.load(ckpt_path, map_location='cpu', weights_only=False)
            _unwrap(model).load_state_dict(checkpoint['model_state'])
            optimizer.load_state_dict(checkpoint['optimizer_state'])
            return checkpoint['epoch'] + 1
    if 'PRETRAINED_CHECKPOINT' in cfg.MODEL and cfg.MODEL.PRETRAINED_CHECKPOINT is not None:
        ckpt_path = cfg.MODEL.PRETRAINED_CHECKPOINT
        if not os.path.exists(ckpt_path):
            print('ERROR: invalid path specified for cfg.MODEL.PRETRAINED_CHECKPOINT')
            print('could not find checkpoint at: ' + ckpt_path)
            exit(-1)
        logger.info(f'Loading pretrained checkpoint at {ckpt_path}')
        checkpoint = torch.load(ckpt_path, map_location='cpu', weights_only=False)
        _unwrap(model).load_state_dict(checkpoint['model_state'])
    return 0
