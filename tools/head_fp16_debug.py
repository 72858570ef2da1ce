"""fp16-forward head chains: which gradient leaves the emulation?  (GPU box)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_head_chain as H

B, nt, T, D, DFF, Hh, L, pad = 2, 6, 20, 256, 256, 4, 1, 3
S = nt * T
g = torch.Generator().manual_seed(11)
x = torch.randn(B, S, D, generator=g); go = torch.randn(B, S, D, generator=g)
mask = torch.ones(B, 1, T); mask[-1, 0, T - pad:] = 0
enc = H._encoder(D, DFF, Hh, L, 0.0, 3)
om = mask.unsqueeze(2).expand(B, 1, nt, T).reshape(B, 1, S)
res = {}
for hd in ('bf16', 'fp16'):
    y, dx, gr = H._run_device(enc, x, mask, go, hd, True)
    yo, dxo, gro = H._run_oracle(enc.cpu(), x, om, go, hd)
    res[hd] = (y, dx, gr, yo, dxo, gro)
    print(hd, 'y', H.rel_l2(y, yo), 'dx', H.rel_l2(dx, dxo))
    for k in sorted(gr):
        print('   %-50s %.3e' % (k, H.rel_l2(gr[k], gro[k])))
print('device fp16 vs device bf16: dx', H.rel_l2(res['fp16'][1], res['bf16'][1]), ' oracle-emu fp16 vs oracle-emu bf16: dx', H.rel_l2(res['fp16'][4], res['bf16'][4]))

# which emulation variant does the device's fp16 mode follow?
from oracle import head as OH
class V(torch.autograd.Function):
    mode = 0
    @staticmethod
    def forward(ctx, x, w, b, f16):
        xf, wf = OH.f16r(x), OH.f16r(w)
        m = V.mode
        xs = OH.bf16r(xf) if m in (0, 1) else xf if m == 2 else OH.bf16r(x)
        ws = OH.bf16r(w) if m in (0, 2, 3) else wf
        ctx.save_for_backward(xs, ws)
        ctx.has_b = b is not None
        y = xf @ wf.t()
        return y + b if b is not None else y
    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        gq = OH.bf16r(dy)
        g2, x2 = gq.reshape(-1, gq.shape[-1]), xr.reshape(-1, xr.shape[-1])
        return gq @ wr, g2.t() @ x2, (g2.sum(0) if ctx.has_b else None), None
orig = OH._EmuLinear
OH._EmuLinear = V
y, dx, gr = res['fp16'][:3]
for m, name in ((0, 'as designed'), (1, 'W of the gradient product = fp16(W)'), (2, 'saved x = fp16(x) unrounded to bf16'), (3, 'saved x = bf16(x)')):
    V.mode = m
    yo, dxo, gro = H._run_oracle(enc.cpu(), x, om, go, 'fp16')
    print('variant %d (%s): dx %.3e fc1.w %.3e fc2.w %.3e' % (m, name, H.rel_l2(dx, dxo), H.rel_l2(gr['enc_layers.0.feed_forward.fc1.weight'], gro['enc_layers.0.feed_forward.fc1.weight']),
          H.rel_l2(gr['enc_layers.0.feed_forward.fc2.weight'], gro['enc_layers.0.feed_forward.fc2.weight'])))

OH._EmuLinear = orig
print('--- ReLU never masks (fc1 bias + 50) ---')
with torch.no_grad():
    for n, q in enc.named_parameters():
        if n.endswith('fc1.bias'):
            q.add_(50.0)
for hd in ('bf16', 'fp16'):
    y, dx, gr = H._run_device(enc, x, mask, go, hd, True)
    yo, dxo, gro = H._run_oracle(enc.cpu(), x, om, go, hd)
    print(hd, 'y', H.rel_l2(y, yo), 'dx', H.rel_l2(dx, dxo), 'fc1.w', H.rel_l2(gr['enc_layers.0.feed_forward.fc1.weight'], gro['enc_layers.0.feed_forward.fc1.weight']))
