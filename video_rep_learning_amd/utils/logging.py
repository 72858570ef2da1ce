"""Rank-gated logging (stdlib only).  Mirrors the behaviour of CARL_MVF/utils/logging.py:40-74 that the
training loop depends on: non-root ranks are silenced, root logs to stdout and to LOGDIR/stdout.log."""
import logging
import os
import sys

_FORMAT = '[%(asctime)s][%(levelname)s] %(name)s: %(lineno)4d: %(message)s'


def _is_root():
    try:
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
    except Exception:  # pragma: no cover
        return True


def setup_logging(output_dir=None):
    root = logging.getLogger()
    root.handlers = []
    if not _is_root():
        root.setLevel(logging.ERROR)
        return
    root.setLevel(logging.INFO)
    fmt = logging.Formatter(_FORMAT, datefmt='%m/%d %H:%M:%S')
    sh = logging.StreamHandler(stream=sys.stdout)
    sh.setFormatter(fmt)
    root.addHandler(sh)
    if output_dir is not None:
        os.makedirs(output_dir, exist_ok=True)
        fh = logging.FileHandler(os.path.join(output_dir, 'stdout.log'))
        fh.setFormatter(fmt)
        root.addHandler(fh)


def get_logger(name):
    return logging.getLogger(name)
