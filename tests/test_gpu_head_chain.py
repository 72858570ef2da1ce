"""Row-chain kernels of the bf16 head (csrc/head_chain.hip) through the C ABI: against the bf16-emulating oracle
(oracle/head.py emulate_head: same restatement, operands rounded where the device rounds them) and against the
one-kernel-per-operator fp32 path of the same model (same dropout masks)."""
import os
import sys
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib, ops  # noqa: E402
from video_rep_learning_amd.models.utils import Encoder  # noqa: E402
from oracle import head as OH  # noqa: E402
from conftest import record_parity  # noqa: E402

DEV = 'cuda'


def rel_l2(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()


def _encoder(D, DFF, H, L, p, seed):
    torch.manual_seed(seed)
    enc = Encoder(D, p, H, DFF, L)
    for n, q in enc.named_parameters():       # biases / LayerNorm affine away from their trivial init values
        if q.dim() == 1:
            with torch.no_grad():
                q.add_(0.1 * torch.randn_like(q))
    return enc


def _run_device(enc, x, mask, go, head_dtype, training, seed=5):
    enc = enc.to(DEV)
    enc.head_dtype = head_dtype
    enc.train(training)
    for q in enc.parameters():
        q.grad = None
    xd = x.to(DEV).requires_grad_(True)
    y = enc(xd, src_mask=None if mask is None else mask.to(DEV), drop_state=ops.DropoutState(seed))
    (y * go.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    return y.detach().cpu(), xd.grad.cpu(), {n: q.grad.detach().cpu().clone() for n, q in enc.named_parameters()}


def _run_oracle(enc, x, mask, go, emulate):
    p = {k: v.detach().double().cpu().requires_grad_(True) for k, v in enc.state_dict().items()}
    xr = x.double().requires_grad_(True)
    OH.emulate_head('bf16' if emulate else None)
    try:
        l0 = enc.enc_layers[0]
        y = OH.encoder(xr, mask, p, '', len(enc.enc_layers), l0.self_att.H, l0.res_layer0.norm.eps)
        (y * go.double()).sum().backward()
    finally:
        OH.emulate_head(None)
    return y.detach(), xr.grad, {k: v.grad for k, v in p.items()}


@pytest.mark.parametrize('B,nt,T,D,DFF,H,L,pad', [(8, 3, 32, 256, 1024, 8, 3, 5), (3, 1, 25, 256, 1024, 8, 2, 0),
                                                   (2, 6, 20, 256, 256, 4, 1, 3), (1, 3, 11, 256, 512, 8, 3, 0)])
def test_encoder_chain_vs_emulating_oracle(B, nt, T, D, DFF, H, L, pad):
    S = nt * T
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, S, D, generator=g)
    go = torch.randn(B, S, D, generator=g)
    mask = torch.ones(B, 1, T)
    if pad:
        mask[-1, 0, T - pad:] = 0
    enc = _encoder(D, DFF, H, L, 0.0, 3)
    assert ops.encoder_chain_supported(D, DFF, H)
    y, dx, gr = _run_device(enc, x, mask, go, 'bf16', True)
    om = mask.unsqueeze(2).expand(B, 1, nt, T).reshape(B, 1, S)
    yo, dxo, gro = _run_oracle(enc.cpu(), x, om, go, True)
    yf, dxf, grf = _run_oracle(enc.cpu(), x, om, go, False)
    # linear_K2d.bias: its true gradient is identically zero (a per-query constant under the softmax)
    keys = [k for k in gr if not k.endswith('linear_K2d.bias')]
    e_y, e_dx = rel_l2(y, yo), rel_l2(dx, dxo)
    worst = max((rel_l2(gr[k], gro[k]), k) for k in keys)
    # what the dtype itself costs (emulating oracle against the fp64 one): the device must sit well inside that
    d_y, d_dx = rel_l2(yo, yf), rel_l2(dxo, dxf)
    d_worst = max((rel_l2(gro[k], grf[k]), k) for k in keys)
    record_parity('head_chain encoder B%d S%d D%d L%d: device vs emulating oracle y %.2e dx %.2e worst param grad %.2e (%s) | '
                  'bf16 vs fp64 oracle y %.2e dx %.2e worst %.2e (%s)' %
                  (B, S, D, L, e_y, e_dx, worst[0], worst[1], d_y, d_dx, d_worst[0], d_worst[1]))
    if L == 1:
        # one layer deep the device and the emulation round the same numbers: agreement far below the dtype's own error
        assert e_y < 2e-4 and e_dx < 2e-4 and worst[0] < 5e-4, (e_y, e_dx, worst)
    else:
        # deeper, the few operands that fall on different sides of a bf16 rounding boundary (fp32 against fp64 arithmetic before
        # the rounding) feed the next layer's roundings: the two drift apart, but stay well inside what the dtype costs
        assert e_y < 0.5 * d_y and e_dx < 0.5 * d_dx and worst[0] < 0.8 * d_worst[0], (e_y, d_y, e_dx, d_dx, worst, d_worst)
    assert d_y < 5e-3 and rel_l2(y, yf) < 5e-3


@pytest.mark.parametrize('p', [0.0, 0.1])
def test_encoder_chain_vs_unfused_fp32_path_same_dropout_masks(p):
    B, nt, T, D, DFF, H, L = 8, 3, 32, 256, 1024, 8, 3
    S = nt * T
    g = torch.Generator().manual_seed(12)
    x = torch.randn(B, S, D, generator=g)
    go = torch.randn(B, S, D, generator=g)
    mask = torch.ones(B, 1, T)
    mask[1, 0, T - 7:] = 0
    enc = _encoder(D, DFF, H, L, p, 4)
    y0, dx0, g0 = _run_device(enc, x, mask, go, 'fp32', True)
    y1, dx1, g1 = _run_device(enc, x, mask, go, 'bf16', True)
    e_y, e_dx = rel_l2(y1, y0), rel_l2(dx1, dx0)
    worst = max((rel_l2(g1[k], g0[k]), k) for k in g0 if not k.endswith('linear_K2d.bias'))
    record_parity('head_chain encoder vs fp32 kernels, dropout %.1f: y %.2e dx %.2e worst param grad %.2e (%s)' %
                  (p, e_y, e_dx, worst[0], worst[1]))
    # bf16 operands (2^-9 relative each) against the fp32 kernels of the same model with the same dropout masks: the dtype's
    # cost as the emulating-oracle test measures it (y 3e-3, gradients 2.5e-2 .. 5e-2), with and without dropout alike
    assert e_y < 6e-3 and e_dx < 4e-2 and worst[0] < 8e-2, (e_y, e_dx, worst)


def test_encoder_chain_gradient_slots_equal_plain_gradients():
    """With the parameters in the flat buffers of the fused optimizer the chain accumulates into the gradient slots (the Q|K|V
    block through its fused view): same numbers as the gradients it returns to autograd otherwise."""
    from video_rep_learning_amd.utils.distributed import FlatBuffers
    B, S, D, DFF, H, L = 4, 48, 256, 1024, 8, 2
    g = torch.Generator().manual_seed(13)
    x = torch.randn(B, S, D, generator=g)
    go = torch.randn(B, S, D, generator=g)
    enc = _encoder(D, DFF, H, L, 0.1, 6)
    _y0, _dx0, g0 = _run_device(enc, x, None, go, 'bf16', True)
    enc = enc.to(DEV)
    groups = [tuple(gr) for m in enc.modules() if callable(getattr(m, 'fuse_groups', None)) for gr in m.fuse_groups()]
    flat = FlatBuffers(list(enc.parameters()), groups)
    owners = [m for m in enc.modules() if callable(getattr(m, 'fuse_groups', None))]
    k = 0
    for m in owners:
        for gr in m.fuse_groups():
            m.set_fused(gr, flat.fused[k])
            k += 1
    flat.zero_grad()
    enc.head_dtype = 'bf16'
    enc.train(True)
    xd = x.to(DEV).requires_grad_(True)
    y = enc(xd, src_mask=None, drop_state=ops.DropoutState(5))
    (y * go.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    for (n, q) in enc.named_parameters():
        got = q._mvf_grad.detach().cpu()
        e = rel_l2(got, g0[n])
        assert e < (1e-5 if 'norm' in n else 1e-6), (n, e)      # LayerNorm gradients are atomically summed
