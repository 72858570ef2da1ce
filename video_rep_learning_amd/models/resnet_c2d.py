"""SimCLR-style projection head, the one class of CARL_MVF/models/resnet_c2d.py that is on the MV-Former path
(MLPHead :112-126).  The ResNet-50 C2D baselines of that file are out of scope."""
import torch.nn as nn

from .. import ops


class MLPHead(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        hidden = cfg.MODEL.PROJECTION_SIZE      # (sic) PROJECTION_HIDDEN_SIZE is ignored by the reference too (:115)
        self.embedding_size = cfg.MODEL.EMBEDDER_MODEL.EMBEDDING_SIZE
        self.net = nn.Sequential(nn.Linear(self.embedding_size, hidden), nn.BatchNorm1d(hidden), nn.ReLU(True),
                                 nn.Linear(hidden, self.embedding_size))
        self.sync_group = None
        # 'bf16': both Linears as row chains (csrc/head_rowlin.hip), BatchNorm and the caller's normalisation inside them
        self.head_dtype = ops.head_dtype_of(cfg)
        self._pack = ops.HeadPack()

    def invalidate_packed(self):
        self._pack.invalidate()

    def chain_active(self):
        lin0, bn, _, lin1 = self.net
        return ops.chain_dtype(self.head_dtype) and bn.momentum is not None and bn.affine and bn.track_running_stats and \
            ops.rowlin_supported(lin0.weight.shape[1], lin0.weight.shape[0]) and ops.rowlin_supported(lin1.weight.shape[1], lin1.weight.shape[0])

    def forward(self, x, normalize=False):
        """normalize: also F.normalize(dim = -1) the result (models/transformer.py:228 does that to this head's output)."""
        b, l, c = x.shape
        lin0, bn, _, lin1 = self.net
        if x.is_cuda and self.chain_active():
            if bn.training != self.training:
                bn.train(self.training)
            from ..utils.distributed import collectives_active
            sync = (self.sync_group,) if isinstance(bn, nn.SyncBatchNorm) and collectives_active() else None
            stages = [ops.RowLinStage(0, 1, bn_out=(bn.running_mean, bn.running_var, bn.momentum), sync=sync),
                      ops.RowLinStage(4, 5, bn_in=(2, 3, bn.eps, True), l2norm=1e-12 if normalize else None)]
            self._pack.set_f16(self.head_dtype == 'fp16')
            y = ops.rowlin_chain(x.reshape(-1, c), stages, [lin0.weight, lin0.bias, bn.weight, bn.bias, lin1.weight, lin1.bias], self.training,
                                 self._pack, (None, (bn.running_mean, bn.running_var)), tag='proj.')
            return y.view(b, l, -1)
        h = ops.linear(x.reshape(-1, c), lin0.weight, lin0.bias)
        h = ops.batch_norm(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training, momentum=bn.momentum,
                           eps=bn.eps, relu=True, sync=isinstance(bn, nn.SyncBatchNorm), group=self.sync_group)
        y = ops.linear(h, lin1.weight, lin1.bias).view(b, l, c)
        return ops.l2_normalize(y) if normalize else y
