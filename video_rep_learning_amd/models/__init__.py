"""Model registry and the checkpoint entry points under the reference's names (`build_model`, `save_checkpoint`,
`load_checkpoint`; CARL_MVF/models/__init__.py).  The I/O itself is utils/checkpoint.py."""
from ..utils import checkpoint as _ckpt
from .transformer import TransformerModel

_EMBEDDERS = {'transformer': TransformerModel}      # MODEL.EMBEDDER_TYPE -> constructor(cfg, local_rank)


def build_model(cfg, local_rank=None):
    kind = cfg.MODEL.EMBEDDER_TYPE
    try:
        ctor = _EMBEDDERS[kind]
    except KeyError:
        raise NotImplementedError("MODEL.EMBEDDER_TYPE '%s': only %s (MV-Former) is on the MI355X path; the ResNet-50 "
                                  "BaseModel of the CARL baselines is out of scope" % (kind, sorted(_EMBEDDERS))) from None
    return ctor(cfg, local_rank)


save_checkpoint = _ckpt.write
load_checkpoint = _ckpt.restore
