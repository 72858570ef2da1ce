"""Command line and config assembly with the reference's surface (CARL_MVF/utils/parser.py): the same flags
(`--local_rank --workdir --logdir --continue_train --visualize --cfg_file --tempcfg --opts K V ...`), a YAML merged
SHALLOWLY over the defaults (a top-level key of the file replaces the default block), `--opts` values typed after the
value they replace, EVAL batch size / frame count tied to TRAIN's, and LOGDIR/config.yml written on first use and re-read
on later runs.  Added for this build: `--local-rank` / LOCAL_RANK (torchrun), `--device`, `--backend`, `--synthetic`,
`--synthetic_raw H W`, `--max_iters`, `--plumbing`, and creation of the optional MI355X.* / MODEL.EMBEDDER_MODEL.* keys from `--opts`."""
import argparse
import os

import yaml

from . import logging
from .config import get_cfg, EasyDict

logger = logging.get_logger(__name__)

# (flags, argparse keywords) -- one row per option
_OPTIONS = (
    (('--local_rank', '--local-rank'), dict(dest='local_rank', type=int, default=None, help='rank in local processes')),
    (('--workdir',), dict(type=str, default='/home/username/datasets', help='Path to datasets and pretrained models.')),
    (('--logdir',), dict(type=str, default=None, help='Path to logs.')),
    (('--continue_train',), dict(action='store_true', default=False)),
    (('--visualize',), dict(action='store_true', default=False)),
    (('--cfg_file',), dict(type=str, default=None, help='Path to the config file')),
    (('--tempcfg',), dict(action='store_true', default=False,
                          help='run with the given config and ignore an existing LOGDIR/config.yml')),
    (('--device',), dict(type=str, default=None, help="'cuda' (MI355X via HIP) -- the only product device")),
    (('--backend',), dict(type=str, default=None, help="torch.distributed backend: 'nccl' (= RCCL) | 'gloo'")),
    (('--synthetic',), dict(action='store_true', default=False, help='train on synthetic clips of the configured shape')),
    (('--synthetic_raw',), dict(type=int, nargs=2, default=None, metavar=('H', 'W'),
                                help='synthetic clips as RAW [0,1] frames of this size: the GPU-side augmentation runs '
                                     'in the loop')),
    (('--max_iters',), dict(type=int, default=0, help='stop each epoch after this many iterations (0 = full)')),
    (('--plumbing',), dict(action='store_true', default=False,
                           help='BASELINE configs[0] without a GPU (--device cpu --backend gloo): run everything around the '
                                'kernels -- config, process group, model and optimizer construction, loader, checkpoint save and '
                                'restore -- and stop at the first HIP call, which raises (there is NO CPU compute path)')),
    (('--opts',), dict(default=None, nargs=argparse.REMAINDER, help='KEY VALUE pairs overriding the config')),
)
# sections whose keys are probed with `in` by the model code and may therefore be introduced from the command line
_OPEN_PREFIXES = ('MI355X.', 'MODEL.EMBEDDER_MODEL.')
_OPEN_KEYS = ('MODEL.BASE_MODEL.WEIGHTS',)


def build_parser():
    p = argparse.ArgumentParser(description='MV-Former SCL training (MI355X-native).')
    for flags, kw in _OPTIONS:
        p.add_argument(*flags, **kw)
    return p


def parse_args(argv=None):
    args = build_parser().parse_args(argv)
    if args.local_rank is None:
        args.local_rank = int(os.environ.get('LOCAL_RANK', 0))
    return args


_BOOL_WORDS = {'true': True, 'True': True, 'false': False, 'False': False}


def convert_value(old, v):
    """`v` (text from the command line) typed like `old`.  bool before int (bool is an int); an unrecognised boolean word
    yields None, as in the reference; list items take the type of the old list's first item and are space-separated
    inside optional brackets."""
    kind = type(old)
    if kind is bool:
        return _BOOL_WORDS.get(v.strip())
    if kind in (str, int, float):
        return kind(v)
    if isinstance(old, (list, tuple)):
        return [convert_value(old[0], item) for item in v.strip('[').strip(']').split(' ')]
    raise ValueError("Don't support for config type:", kind)


def _override(cfg, dotted, text):
    *parents, leaf = dotted.split('.')
    if parents and parents[0] == 'MI355X' and 'MI355X' not in cfg:      # the build's own optional section
        cfg['MI355X'] = EasyDict()
    node = cfg
    for name in parents:
        node = node[name]                                               # KeyError for an unknown section, as upstream
    if leaf in node:
        node[leaf] = convert_value(node[leaf], text)
    elif dotted.startswith(_OPEN_PREFIXES) or dotted in _OPEN_KEYS:
        node[leaf] = yaml.safe_load(text)
    else:
        raise KeyError(dotted)


def load_config(args):
    cfg = get_cfg()
    path = getattr(args, 'cfg_file', None)
    if path is not None and os.path.exists(path):
        logger.info('Using config from %s.', path)
        with open(path, 'r') as f:
            cfg.update(yaml.safe_load(f))
    pairs = getattr(args, 'opts', None) or ()
    for key, text in zip(pairs[0::2], pairs[1::2]):
        _override(cfg, key, text)
    logdir = getattr(args, 'logdir', None)
    cfg.LOGDIR = logdir if logdir is not None else os.path.join('/tmp', cfg.LOGDIR)
    cfg.EVAL.BATCH_SIZE = cfg.TRAIN.BATCH_SIZE
    cfg.EVAL.NUM_FRAMES = cfg.TRAIN.NUM_FRAMES
    return cfg


def to_dict(config):
    """EasyDict tree -> plain containers (what yaml.safe_dump accepts)."""
    if isinstance(config, dict):
        return {k: to_dict(v) for k, v in config.items()}
    if isinstance(config, (list, tuple)):
        return [to_dict(c) for c in config]
    return config


def setup_train_dir(cfg, logdir, continue_train=False, tempcfg=False):
    """First run in `logdir`: store the effective config as config.yml.  Later runs: the stored file wins over the command
    line (so a resumed job cannot silently change shape) unless --tempcfg says otherwise."""
    os.makedirs(os.path.join(logdir, 'train_logs'), exist_ok=True)
    stored = os.path.join(logdir, 'config.yml')
    if not os.path.exists(stored):
        logger.info('Using config from config.py as no config.yml file exists in %s', logdir)
        with open(stored, 'w') as f:
            yaml.safe_dump(to_dict({k: v for k, v in cfg.items() if k != 'args'}), f, default_flow_style=False)
        return
    if tempcfg:
        print('tempcfg mode enabled, will ignore existing config file')
        return
    logger.info('Using config from config.yml that exists in %s.', logdir)
    with open(stored, 'r') as f:
        cfg.update(yaml.safe_load(f))
