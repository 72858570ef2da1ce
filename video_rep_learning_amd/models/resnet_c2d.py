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

    def forward(self, x):
        b, l, c = x.shape
        lin0, bn, _, lin1 = self.net
        h = ops.linear(x.reshape(-1, c), lin0.weight, lin0.bias)
        h = ops.batch_norm(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training, momentum=bn.momentum,
                           eps=bn.eps, relu=True, sync=isinstance(bn, nn.SyncBatchNorm), group=self.sync_group)
        return ops.linear(h, lin1.weight, lin1.bias).view(b, l, c)
