"""`get_algo(cfg)` -> object with `compute_loss(...)` (the reference's algo registry surface).  TRAINING_ALGO 'scl' is the
algorithm of every configs_mvf/*.yml; tcc / tcn / classification are the CARL baselines and not on the MI355X path."""
from . import scl as _scl

_REGISTRY = {}


def register(name):
    def deco(cls):
        _REGISTRY[name] = cls
        return cls
    return deco


register('scl')(_scl.SCL)


def get_algo(cfg):
    name = cfg.TRAINING_ALGO
    cls = _REGISTRY.get(name)
    if cls is None:
        raise ValueError('%s not supported yet.' % name)     # the reference's message for an unknown algo
    return cls(cfg)
