"""fp16-forward head chains: the saved ReLU activation `a` of the encoder layer against the emulation's pre-activation (GPU box)"""
import os, sys, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_head_chain as H
from video_rep_learning_amd import ops
from oracle import head as OH
hip = ctypes.CDLL('libamdhip64.so')

B, nt, T, D, DFF, Hh, L, pad = 2, 6, 20, 256, 256, 4, 1, 3
S = nt * T
g = torch.Generator().manual_seed(11)
x = torch.randn(B, S, D, generator=g); go = torch.randn(B, S, D, generator=g)
mask = torch.ones(B, 1, T); mask[-1, 0, T - pad:] = 0
enc = H._encoder(D, DFF, Hh, L, 0.0, 3)
cap = {}
real = ops.call
def spy(name, *args):
    rc = real(name, *args)
    if name == 'mvf_enc_layer_fwd':
        s = args[0]._obj
        if s.a:
            cap['a'] = (int(s.a), s.M, s.DFF)
    return rc
ops.call = spy
for hd in ('bf16', 'fp16'):
    y, dx, gr = H._run_device(enc, x, mask, go, hd, True)
    p, M, F = cap['a']
    host = (ctypes.c_uint16 * (M * F))()
    hip.hipMemcpy(host, ctypes.c_void_p(p), M * F * 2, 2)
    bits = torch.frombuffer(host, dtype=torch.int16).view(M, F).clone()
    a = bits.view(torch.float16 if hd == 'fp16' else torch.bfloat16).double()
    cap[hd] = a
    print(hd, 'saved a: rows', M, 'cols', F, 'positive fraction %.4f' % (a > 0).double().mean().item(), 'max', a.max().item(), 'nan', torch.isnan(a).sum().item())
print('mask disagreement bf16-run vs fp16-run: %.5f' % ((cap['bf16'] > 0) != (cap['fp16'] > 0)).double().mean().item())
d = (cap['bf16'] - cap['fp16']).abs()
print('value disagreement: max %.3e mean %.3e' % (d.max().item(), d.mean().item()))
bad = ((cap['bf16'] > 0) != (cap['fp16'] > 0)).nonzero()
print('first disagreements (row, col, bf16 a, fp16 a):', [(int(r), int(c), cap['bf16'][r, c].item(), cap['fp16'][r, c].item()) for r, c in bad[:12]])

# the emulation's pre-activation of fc1 in both modes
om = mask.unsqueeze(2).expand(B, 1, nt, T).reshape(B, 1, S)
zs = {}
real_lin = OH.linear
def spy_lin(xx, p, name):
    yy = real_lin(xx, p, name)
    if name.endswith('fc1'):
        zs[cur] = yy.detach().reshape(-1, yy.shape[-1]).clone()
    return yy
OH.linear = spy_lin
for cur in ('bf16', 'fp16', False):
    H._run_oracle(enc.cpu(), x, om, go, cur)
for hd in ('bf16', 'fp16'):
    z = zs[hd]
    dis = ((z > 0) != (cap[hd] > 0))
    print('%s: device mask vs emulation mask disagreement %.6f (of %d); |z_emu| at the disagreements: %s' % (
        hd, dis.double().mean().item(), dis.numel(), [float('%.2e' % v) for v in z[dis].abs()[:8].tolist()]))
    err = (torch.relu(z) - cap[hd]).abs()
    print('   relu(z_emu) vs saved a: max abs %.3e' % err.max().item())

# fc1.bias gradient three ways (fp16-forward run): device, emulation, host arithmetic on the DEVICE's saved a
OH.linear = real_lin
y, dx, gr = H._run_device(enc, x, mask, go, 'fp16', True)
yo, dxo, gro = H._run_oracle(enc.cpu(), x, om, go, 'fp16')
W2 = enc.enc_layers[0].feed_forward.fc2.weight.detach().double().cpu()      # [D, DFF]
g2 = OH.bf16r(go.double().reshape(-1, D))
du = (g2 @ OH.bf16r(W2)) * (cap['fp16'] > 0)
hb = OH.bf16r(du).sum(0)
k = 'enc_layers.0.feed_forward.fc1.bias'
print('fc1.bias grad: device vs emulation %.3e, device vs host(a_dev mask) %.3e, emulation vs host %.3e' % (
    H.rel_l2(gr[k], gro[k]), H.rel_l2(gr[k], hb), H.rel_l2(gro[k], hb)))
