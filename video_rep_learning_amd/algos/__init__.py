"""Training-algorithm registry (CARL_MVF/algos/__init__.py:7-20).  Only 'scl' -- the algorithm of every
configs_mvf/*.yml -- is on the MI355X hot path; tcc/tcn/classification are the original CARL baselines."""
from .scl import SCL

ALGO_NAME_TO_ALGO_CLASS = {
    'scl': SCL,
}


def get_algo(cfg):
    algo_name = cfg.TRAINING_ALGO
    if algo_name not in ALGO_NAME_TO_ALGO_CLASS:
        raise ValueError('%s not supported yet.' % algo_name)
    return ALGO_NAME_TO_ALGO_CLASS[algo_name](cfg)
