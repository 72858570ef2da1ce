"""Temporal-transformer building blocks with the reference's class / parameter names
(CARL_MVF/models/utils.py:47-242) so `embed.video_encoder.*` checkpoints load unchanged; every forward is a
chain of HIP ops (video_rep_learning_amd.ops): LayerNorm -> fused QKV GEMM -> attention -> out-proj GEMM ->
dropout+residual -> LayerNorm -> FC1+ReLU GEMM -> FC2 GEMM -> dropout+residual."""
import math
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn

from .. import ops


def generate_sincos_embedding(seq_len, d_model, train_len=None):
    """utils.py:113-126 (vectorised; same float64 numbers).  Even channel i: sin(pos / 10000^(i/d)), odd
    channel i: cos(pos / 10000^(i/d)) -- the exponent uses the channel index itself."""
    pos = np.arange(seq_len, dtype=np.float64) if train_len is None else np.linspace(0, train_len - 1, num=seq_len)
    ch = np.arange(d_model, dtype=np.float64)
    ang = pos[:, None] / (10000.0 ** (ch[None, :] / d_model))
    tab = np.where((np.arange(d_model) % 2 == 0)[None, :], np.sin(ang), np.cos(ang))
    return torch.from_numpy(tab).unsqueeze(0)


class PositionalEncoder(nn.Module):
    """utils.py:128-145.  The table is built once per (S, d, device) and kept on the device (the reference
    rebuilds it in a Python loop and copies it host->device every forward).  The add itself is fused into the
    epilogue of the preceding `video_emb` GEMM (see MultiEntityTransformerEmbModel), `forward` is the plain form."""

    def __init__(self, cfg, d_model, dout_p, seq_len=3660):
        super().__init__()
        self.cfg = cfg
        self.d_model = d_model
        self.dout_p = dout_p
        self.seq_len = seq_len
        self._tables = {}

    def table(self, S, device):
        key = (S, str(device))
        if key not in self._tables:
            t = generate_sincos_embedding(S, self.d_model, None if S == self.seq_len else self.seq_len)
            self._tables[key] = t[0].float().to(device).contiguous()
        return self._tables[key]


class MultiheadedAttention(nn.Module):
    def __init__(self, d_model_Q, d_model_K, d_model_V, H, dout_p=0.0, d_model=None, d_out=None):
        super().__init__()
        self.H = H
        self.d_model = d_model if d_model is not None else d_model_Q
        self.d_out = d_out if d_out is not None else d_model_Q
        self.d_k = self.d_model // H
        assert self.d_model % H == 0
        self.linear_Q2d = nn.Linear(d_model_Q, self.d_model)
        self.linear_K2d = nn.Linear(d_model_K, self.d_model)
        self.linear_V2d = nn.Linear(d_model_V, self.d_model)
        self.linear_d2Q = nn.Linear(self.d_model, self.d_out)

    # ---- Q|K|V as one GEMM operand.  With the flat parameter buffers of the fused optimizer the three weights (and
    # the three biases) are laid out back to back (fuse_groups), so a VIEW over them is the fused [3d, d_in] weight and
    # its gradient block is one contiguous slot; without an optimizer (eval, plain autograd tests) they are concatenated.
    def fuse_groups(self):
        return [(self.linear_Q2d.weight, self.linear_K2d.weight, self.linear_V2d.weight),
                (self.linear_Q2d.bias, self.linear_K2d.bias, self.linear_V2d.bias)]

    def set_fused(self, group, views):
        if not hasattr(self, '_fused'):
            self._fused = {}
        self._fused['w' if group[0] is self.linear_Q2d.weight else 'b'] = views

    def _qkv_operands(self):
        f = getattr(self, '_fused', None)
        if f and f.get('w') is not None and f.get('b') is not None and \
                f['w'][0].data_ptr() == self.linear_Q2d.weight.data_ptr() and \
                f['b'][0].data_ptr() == self.linear_Q2d.bias.data_ptr():
            owners = [p for g in self.fuse_groups() for p in g]
            return f['w'][0], f['b'][0], (f['w'][1], f['b'][1], owners)
        w = torch.cat([self.linear_Q2d.weight, self.linear_K2d.weight, self.linear_V2d.weight], 0)
        b = torch.cat([self.linear_Q2d.bias, self.linear_K2d.bias, self.linear_V2d.bias], 0)
        return w, b, None

    def forward(self, x, mask=None, resid=None, drop=None):
        """Self-attention only (Q = K = V = x), x [B, S, D], mask [B, 1, S] (or [B, 1, S / k], read periodically: the frame mask
        of a joint entity x frame sequence); `resid`/`drop`: the residual connection
        and its dropout, fused into the output projection's epilogue."""
        B, S, _ = x.shape
        w, b, fused = self._qkv_operands()
        qkv = ops.linear(x.reshape(B * S, -1), w, b, fused=fused)
        o = ops.temporal_attention(qkv, None if mask is None else mask.reshape(B, -1), B, S, self.H)     # [B, S] or periodic
        return ops.linear(o, self.linear_d2Q.weight, self.linear_d2Q.bias,
                          resid=None if resid is None else resid.reshape(B * S, -1), drop=drop).view(B, S, -1)


class ResidualConnection(nn.Module):
    def __init__(self, size, dout_p):
        super().__init__()
        self.norm = nn.LayerNorm(size)
        self.dout_p = dout_p


class PositionwiseFeedForward(nn.Module):
    def __init__(self, d_model, d_ff, dout_p):
        super().__init__()
        self.fc1 = nn.Linear(d_model, d_ff)
        self.fc2 = nn.Linear(d_ff, d_model)

    def forward(self, x, resid=None, drop=None):
        return ops.linear(ops.linear(x, self.fc1.weight, self.fc1.bias, relu=True), self.fc2.weight, self.fc2.bias,
                          resid=resid, drop=drop)


class EncoderLayer(nn.Module):
    def __init__(self, d_model, dout_p, H=8, d_ff=None, d_hidden=None):
        super().__init__()
        self.res_layer0 = ResidualConnection(d_model, dout_p)
        self.res_layer1 = ResidualConnection(d_model, dout_p)
        d_hidden = d_model if d_hidden is None else d_hidden
        d_ff = 4 * d_model if d_ff is None else d_ff
        self.self_att = MultiheadedAttention(d_model, d_model, d_model, H, d_model=d_hidden)
        self.feed_forward = PositionwiseFeedForward(d_model, d_ff, dout_p=0.0)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, x, src_mask=None, drop_state=None):
        r0, r1 = self.res_layer0, self.res_layer1
        # x + drop(sub(LN(x))) twice; the add and the dropout live in the epilogue of each sub-layer's last GEMM, and the two
        # gradients of x (residual path + LayerNorm) are summed inside the LayerNorm backward kernel (ops.layer_norm_fork)
        x, h = ops.layer_norm_fork(x, r0.norm.weight, r0.norm.bias, r0.norm.eps)
        x = self.self_att(h, src_mask, resid=x, drop=ops.drop_args(r0.dout_p, self.training, drop_state, x.numel()))
        x, h = ops.layer_norm_fork(x, r1.norm.weight, r1.norm.bias, r1.norm.eps)
        return self.feed_forward(h, resid=x, drop=ops.drop_args(r1.dout_p, self.training, drop_state, x.numel()))


class Encoder(nn.Module):
    def __init__(self, d_model, dout_p, H, d_ff, N, d_hidden=None):
        super().__init__()
        layer = EncoderLayer(d_model, dout_p, H, d_ff, d_hidden)
        self.enc_layers = nn.ModuleList([deepcopy(layer) for _ in range(N)])

        # 'bf16': run on the row-chain kernels (csrc/head_chain.hip) where the shapes allow it; set from cfg by the owning model
        # (ops.head_dtype_of); 'fp32' = one kernel per operator on the fp32 matrix cores (parity mode)
        self.head_dtype = 'fp32'
        self._pack = ops.HeadPack()

    def invalidate_packed(self):
        """The fused optimizer updated the parameters through raw pointers: the bf16 operand copies are stale."""
        self._pack.invalidate()

    def _chain_layers(self):
        layers = []
        for ly in self.enc_layers:
            att, ff = ly.self_att, ly.feed_forward
            if att.d_model != att.linear_Q2d.weight.shape[1] or att.d_out != att.d_model:
                return None
            w, b, fused = att._qkv_operands()
            plist = [ly.res_layer0.norm.weight, ly.res_layer0.norm.bias, w, b, att.linear_d2Q.weight, att.linear_d2Q.bias,
                     ly.res_layer1.norm.weight, ly.res_layer1.norm.bias, ff.fc1.weight, ff.fc1.bias, ff.fc2.weight, ff.fc2.bias]
            slots = [ops.grad_slot(q) for q in plist]
            owners = [q for i, q in enumerate(plist) if i not in (2, 3)]
            if fused is not None:
                slots[2], slots[3] = fused[0], fused[1]
                owners += list(fused[2])
            layers.append({'params': plist, 'slots': slots if all(s_ is not None for s_ in slots) else None, 'owners': owners})
        return layers

    def chain_active(self):
        """True when forward() takes the row-chain kernels (bf16 operands) for this module's shapes."""
        l0 = self.enc_layers[0] if len(self.enc_layers) > 0 else None
        if not ops.chain_dtype(self.head_dtype) or l0 is None:
            return False
        att = l0.self_att
        return att.d_model == att.linear_Q2d.weight.shape[1] and att.d_out == att.d_model and \
            ops.encoder_chain_supported(att.d_model, l0.feed_forward.fc1.weight.shape[0], att.H)

    def forward(self, x, src_mask=None, drop_state=None):
        l0 = self.enc_layers[0] if len(self.enc_layers) > 0 else None
        if x.is_cuda and self.chain_active():
            layers = self._chain_layers()
            if layers is not None:
                drops = []
                for ly in self.enc_layers:
                    drops.append(ops.drop_args(ly.res_layer0.dout_p, self.training, drop_state, x.numel()))
                    drops.append(ops.drop_args(ly.res_layer1.dout_p, self.training, drop_state, x.numel()))
                self._pack.set_f16(self.head_dtype == 'fp16')
                return ops.encoder_chain(x, src_mask, layers, l0.self_att.H, l0.res_layer0.norm.eps, drops, self._pack)
        for layer in self.enc_layers:
            x = layer(x, src_mask, drop_state)
        return x
