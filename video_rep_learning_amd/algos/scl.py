"""Sequence-contrastive loss algorithm with the reference's interface (CARL_MVF/algos/scl.py:18-105):
`SCL(cfg).compute_loss(model, videos, seq_lens, chosen_steps, video_masks, training) -> {"loss": 0-dim tensor}`.
`compute_sequence_loss` is one fused HIP forward + one fused HIP backward (csrc/scl_loss.hip) instead of six
dense [M,M] temporaries and Python loops over the batch.

New capability (north star): with cfg.MI355X.GATHER_EMBEDDINGS and a 'batch*' NEGATIVE_TYPE the
embeddings (+ steps / seq_lens / masks) of all ranks are all-gathered (RCCL) so that every rank contrasts
against W*B videos; the gradient of the global loss w.r.t. the local rows needs no further communication.  With
a 'single*' NEGATIVE_TYPE cross-video negatives carry zero weight, the gather is a mathematical no-op and is
skipped."""
import torch

from .. import ops
from ..utils import distributed as du


class SCL(object):
    def __init__(self, cfg):
        self.cfg = cfg
        self.positive_type = cfg.SCL.POSITIVE_TYPE
        self.negative_type = cfg.SCL.NEGATIVE_TYPE
        self.temperature = cfg.SCL.SOFTMAX_TEMPERATURE
        self.label_varience = cfg.SCL.LABEL_VARIENCE
        self.embedding_size = cfg.MODEL.EMBEDDER_MODEL.EMBEDDING_SIZE
        self.positive_window = cfg.SCL.POSITIVE_WINDOW
        if self.positive_type != 'gauss':
            raise NotImplementedError("SCL.POSITIVE_TYPE '%s': the reference produces an all-zero label (zero loss) "
                                      "for anything but 'gauss'" % self.positive_type)
        mi = cfg.MI355X if 'MI355X' in cfg else {}
        self.gather = bool(mi['GATHER_EMBEDDINGS']) if 'GATHER_EMBEDDINGS' in mi else False

    def compute_loss(self, model, videos, seq_lens, chosen_steps, video_masks=None, training=True):
        num_frames = self.cfg.TRAIN.NUM_FRAMES
        batch_size, num_views, num_steps, c, h, w = videos.shape
        videos = videos.view(-1, num_steps, c, h, w)
        if video_masks is not None:
            video_masks = video_masks.view(-1, 1, num_steps)
        embs = model(videos, num_frames, video_masks=video_masks, project=self.cfg.MODEL.PROJECTION)
        embs = embs.view(batch_size, num_views, num_frames, embs.size(-1))
        seq_lens = seq_lens.view(batch_size, num_views)
        dev = embs.device
        return self.compute_sequence_loss(embs, seq_lens.to(dev), chosen_steps.to(dev), video_masks.to(dev))

    def compute_sequence_loss(self, embs, seq_lens, steps, masks=None):
        batch_size, num_views, num_frames, channels = embs.shape
        assert num_views == 2
        m_local = batch_size * num_views * num_frames
        e = embs.reshape(m_local, channels)
        per_row = ops.scl_rows(steps, seq_lens, masks) if masks is not None else None     # one launch for the three vectors
        if per_row is not None:
            st, ln, mk = per_row[0], per_row[1], per_row[2]
        else:
            st = steps.reshape(m_local).float()
            ln = seq_lens.reshape(batch_size, num_views, 1).expand(batch_size, num_views, num_frames).reshape(m_local).float()
            mk = masks.reshape(m_local).float()
        row0, rows, scale = 0, None, 1.0
        if self.gather and 'single' not in self.negative_type and du.collectives_active():
            ws, rk = du.get_world_size(), du.get_rank()
            e = du.gather_rows(e)
            st, ln, mk = du.all_gather([st, ln, mk])
            # every rank holds the same global loss; DDP averages gradients over ranks, so scale the local
            # slice by W to end up with the true gradient of the global loss
            row0, rows, scale = rk * m_local, m_local, float(ws)
        loss = ops.scl_loss(e, st, ln, mk, num_frames, self.negative_type, self.temperature, self.label_varience,
                            row0=row0, rows=rows, grad_scale=scale)
        return {'loss': loss}
