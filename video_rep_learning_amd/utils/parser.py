"""CLI + config loading with the reference's surface (CARL_MVF/utils/parser.py:15-131):
`--local_rank --workdir --logdir --continue_train --visualize --cfg_file --opts K V ... --tempcfg`, YAML merged
SHALLOWLY over the defaults, `--opts` values typed by the existing value, EVAL batch/frames forced to TRAIN's.
Additions (torchrun era): `--local-rank` alias and the LOCAL_RANK env default; `--device`/`--backend`."""
import argparse
import os

import yaml

from . import logging
from .config import get_cfg, EasyDict

logger = logging.get_logger(__name__)


def build_parser():
    p = argparse.ArgumentParser(description='MV-Former SCL training (MI355X-native).')
    p.add_argument('--local_rank', '--local-rank', dest='local_rank', type=int,
                   default=int(os.environ.get('LOCAL_RANK', 0)), help='rank in local processes')
    p.add_argument('--workdir', type=str, default='/home/username/datasets', help='Path to datasets and pretrained models.')
    p.add_argument('--logdir', type=str, default=None, help='Path to logs.')
    p.add_argument('--continue_train', action='store_true', default=False)
    p.add_argument('--visualize', action='store_true', default=False)
    p.add_argument('--cfg_file', type=str, default=None, help='Path to the config file')
    p.add_argument('--tempcfg', action='store_true', default=False,
                   help='run with the given config and ignore an existing LOGDIR/config.yml')
    p.add_argument('--device', type=str, default=None, help="'cuda' (MI355X via HIP) -- the only product device")
    p.add_argument('--backend', type=str, default=None, help="torch.distributed backend: 'nccl' (= RCCL) | 'gloo'")
    p.add_argument('--synthetic', action='store_true', default=False, help='train on synthetic clips of the configured shape')
    p.add_argument('--synthetic_raw', type=int, nargs=2, default=None, metavar=('H', 'W'),
                   help='synthetic clips as RAW [0,1] frames of this size: the GPU-side augmentation runs in the loop')
    p.add_argument('--max_iters', type=int, default=0, help='stop each epoch after this many iterations (0 = full)')
    p.add_argument('--opts', default=None, nargs=argparse.REMAINDER, help='KEY VALUE pairs overriding the config')
    return p


def parse_args(argv=None):
    return build_parser().parse_args(argv)


def convert_value(old, v):
    """Type `v` (a string from the command line) like the existing value (parser.py:46-61)."""
    if isinstance(old, bool):
        s = v.strip()
        if s in ('False', 'false'):
            return False
        if s in ('True', 'true'):
            return True
        return None  # the reference falls through and returns None here
    if isinstance(old, str):
        return str(v)
    if isinstance(old, int):
        return int(v)
    if isinstance(old, float):
        return float(v)
    if isinstance(old, (list, tuple)):
        return [convert_value(old[0], x) for x in v.strip('[').strip(']').split(' ')]
    raise ValueError("Don't support for config type:", type(old))


def load_config(args):
    cfg = get_cfg()
    if getattr(args, 'cfg_file', None) is not None and os.path.exists(args.cfg_file):
        logger.info('Using config from %s.', args.cfg_file)
        with open(args.cfg_file, 'r') as f:
            cfg.update(yaml.safe_load(f))
    opts = getattr(args, 'opts', None)
    if opts:
        for full_key, v in zip(opts[0::2], opts[1::2]):
            keys = full_key.split('.')
            d = cfg
            if keys[0] == 'MI355X' and 'MI355X' not in cfg:      # the build's own optional section
                cfg['MI355X'] = EasyDict()
            for k in keys[:-1]:
                d = d[k]
            # reference: d[subkey] must exist (KeyError otherwise); build-specific optional keys may be created
            if keys[-1] in d:
                d[keys[-1]] = convert_value(d[keys[-1]], v)
            elif keys[0] == 'MI355X' or full_key.startswith('MODEL.EMBEDDER_MODEL.') or full_key == 'MODEL.BASE_MODEL.WEIGHTS':
                d[keys[-1]] = yaml.safe_load(v)
            else:
                raise KeyError(full_key)
    if getattr(args, 'logdir', None) is not None:
        cfg.LOGDIR = args.logdir
    else:
        cfg.LOGDIR = os.path.join('/tmp', cfg.LOGDIR)
    cfg.EVAL.BATCH_SIZE = cfg.TRAIN.BATCH_SIZE
    cfg.EVAL.NUM_FRAMES = cfg.TRAIN.NUM_FRAMES
    return cfg


def to_dict(config):
    if isinstance(config, (list, tuple)):
        return [to_dict(c) for c in config]
    if isinstance(config, dict):
        return {k: to_dict(v) for k, v in config.items()}
    return config


def setup_train_dir(cfg, logdir, continue_train=False, tempcfg=False):
    """parser.py:106-131: persist the config on first use, otherwise re-read the stored one (unless --tempcfg)."""
    os.makedirs(logdir, exist_ok=True)
    config_path = os.path.join(logdir, 'config.yml')
    if not os.path.exists(config_path):
        logger.info('Using config from config.py as no config.yml file exists in %s', logdir)
        with open(config_path, 'w') as f:
            yaml.safe_dump({k: to_dict(v) for k, v in cfg.items() if k != 'args'}, f, default_flow_style=False)
    elif tempcfg:
        print('tempcfg mode enabled, will ignore existing config file')
    else:
        logger.info('Using config from config.yml that exists in %s.', logdir)
        with open(config_path, 'r') as f:
            cfg.update(yaml.safe_load(f))
    os.makedirs(os.path.join(logdir, 'train_logs'), exist_ok=True)
