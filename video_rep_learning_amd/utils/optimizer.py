"""Optimizer / LR-schedule construction with the reference's surface (CARL_MVF/utils/optimizer.py:10-117).

`construct_optimizer` selects the same parameters in the same two groups (BN / non-BN; backbone skipped when
MODEL.TRAIN_BASE == 'frozen').  For Adam -- the optimizer of every shipped config -- it returns `FusedAdam`: a
`torch.optim.Optimizer` (so torch LR schedulers and `state_dict()` consumers keep working, checkpoints stay in
torch.optim.Adam's format) whose parameters/gradients/moments live in flat buffers and whose step, including the
global-norm clip of train.py:124-126, is the HIP kernel pair of csrc/optim.hip."""
import math

import numpy as np
import torch

from .. import ops
from .distributed import FlatBuffers, GradReducer


def select_parameters(model, cfg):
    """(bn_params, non_bn_params) exactly as optimizer.py:26-42."""
    bn, non_bn = [], []
    for n, m in model.named_modules():
        is_bn = isinstance(m, torch.nn.modules.batchnorm._NormBase) or getattr(m, '_is_batchnorm', False)
        for p in m.parameters(recurse=False):
            if not p.requires_grad:
                continue
            if 'backbone' in n and cfg.MODEL.TRAIN_BASE != 'train_all':
                if cfg.MODEL.TRAIN_BASE == 'frozen':
                    continue
                if cfg.MODEL.TRAIN_BASE == 'only_bn' and is_bn:
                    bn.append(p)
            else:
                (bn if is_bn else non_bn).append(p)
    return bn, non_bn


def find_fuse_groups(model):
    """Parameter tuples that a module wants laid out back to back in the flat buffers (module.fuse_groups())."""
    groups = []
    for m in model.modules():
        fg = getattr(m, 'fuse_groups', None)
        if callable(fg):
            for g in fg():
                groups.append((m, tuple(g)))
    return groups


class FusedAdam(torch.optim.Optimizer):
    """Adam with L2 weight decay (== torch.optim.Adam(weight_decay=wd)) on flat buffers + fused global-norm clip.

    Two deliberate differences from torch.optim.Adam behind DDP(find_unused_parameters=True):
      * a step whose (clipped) gradient norm is NaN / Inf is skipped ON THE DEVICE -- parameters, moments and the bias-correction
        step count stay as they were (what GradScaler.step does on the reference's fp16 path, train.py:127-133); the count of
        skipped steps is `skipped_steps()` (one host sync) and is taken off `step` in `state_dict()`.  The norm (and with it
        the guard) is computed every step, with or without clipping;
      * every parameter of the flat buffer is updated every step.  torch skips a parameter whose .grad is None; here a parameter
        that received no gradient has a ZERO slot, so it still sees weight decay and moment decay.  Every trainable parameter of
        the MV-Former configs receives a gradient each step except `embed.pooling.cross_att.linear_K2d.bias`, whose gradient is
        identically zero in the reference too (a per-query constant under the softmax), so the two only differ there by
        rounding-noise-sized gradients (tests list it among the null-gradient tensors)."""

    def __init__(self, param_groups, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, bucket_bytes=8 << 20,
                 fuse_groups=()):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(param_groups, defaults)
        params = [p for g in self.param_groups for p in g['params']]
        self.flat = FlatBuffers(params, [g for _m, g in fuse_groups])
        for (m, g), views in zip([fg for fg in fuse_groups if all(any(q is p for p in params) for q in fg[1])], self.flat.fused):
            m.set_fused(g, views)
        self.exp_avg = torch.zeros_like(self.flat.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat.flat_p)
        self.step_count = 0
        dev = self.flat.flat_p.device
        self._scratch = torch.empty(1024, device=dev, dtype=torch.float32)
        self._norm = torch.zeros(2, device=dev, dtype=torch.float32)     # [gradient norm, skipped (non-finite) steps so far]
        self.reducer = GradReducer(self.flat, bucket_bytes)
        # modules that keep a packed device copy of parameters this optimizer updates through raw pointers (torch's version
        # counters do not see the kernel's writes): told after every step (models/vit.VisionTransformer.invalidate_packed)
        self.packed_owners = []
        # element range of every param group (groups are contiguous in the flat buffer by construction: fuse groups
        # only permute parameters inside one param group)
        self._ranges = []
        k = 0
        for g in self.param_groups:
            n = len(g['params'])
            if n == 0:
                self._ranges.append((0, 0))
            else:
                offs = [(self.flat.offsets[i], (self.flat.params[i].numel() + 3) // 4 * 4) for i in range(k, k + n)]
                s0, e0 = min(o for o, _ in offs), max(o + m for o, m in offs)
                assert e0 - s0 == sum(m for _, m in offs), 'a fuse group straddles two optimizer param groups'
                self._ranges.append((s0, e0))
            k += n
        # the Adam kernel doubles as zero_grad(): it must see every element of the gradient buffer
        assert sum(e - s_ for s_, e in self._ranges) == self.flat.numel, 'optimizer param groups do not cover the flat buffers'

    def zero_grad(self, set_to_none=False):
        """Free right after a step (the Adam kernel zeroed the buffer while reading it); a real fill only when gradients
        were written since (FlatBuffers.dirty: a backward without a step, manual writes through ops.grad_slot)."""
        self.flat.zero_grad()

    @torch.no_grad()
    def step(self, closure=None, max_norm=0.0):
        """Waits for the gradient all-reduce, then clip (if max_norm > 0) + Adam in place. Returns the device
        scalar holding the (averaged) gradient norm when clipping, else None.  The norm is computed in either case: it
        is also the finite-gradient guard of the update (6 us on 4.8 M parameters)."""
        gscale = self.reducer.finish()
        self.step_count += 1
        norm = ops.grad_norm(self.flat.flat_g, self._scratch, self._norm)
        # adjacent param groups with the same hyper-parameters (the reference's BN / non-BN groups always have: optimizer.py:44-52)
        # are one launch
        runs = []
        for g, (s, e) in zip(self.param_groups, self._ranges):
            if e <= s:
                continue
            hp = (float(g['lr']), g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'])
            if runs and runs[-1][0] == hp and runs[-1][2] == s:
                runs[-1][2] = e
            else:
                runs.append([hp, s, e])
        for hp, s, e in runs:
            ops.adam_step(self.flat.flat_p[s:e], self.flat.flat_g[s:e], self.exp_avg[s:e], self.exp_avg_sq[s:e],
                          hp[0], hp[1], hp[2], hp[3], hp[4], self.step_count,
                          clip=float(max_norm or 0.0), norm=norm, gscale=gscale, zero_grad=True)
        self.flat.dirty = False           # every element of the gradient buffer lies in one of the ranges (asserted above)
        for m in self.packed_owners:
            m.invalidate_packed()
        return norm[:1] if (max_norm and max_norm > 0) else None

    def skipped_steps(self):
        """Steps dropped because their gradient norm was not finite (device counter; reading it synchronises)."""
        return int(self._norm[1].item())

    # ---- torch.optim.Adam-compatible (de)serialisation: CARL_MVF/models/__init__.py:22-27,42-46 ----
    def state_dict(self):
        """Side-effect free: the saved `step` is the number of steps that really updated the parameters (skipped ones taken
        off, one host sync); the optimizer's own counters are left as they are."""
        state = {}
        steps = max(self.step_count - self.skipped_steps(), 0) if self.step_count > 0 else 0
        for i, (p, o) in enumerate(zip(self.flat.params, self.flat.offsets)):
            n = p.numel()
            state[i] = {'step': torch.tensor(float(steps)),
                        'exp_avg': self.exp_avg[o:o + n].view_as(p).clone(),
                        'exp_avg_sq': self.exp_avg_sq[o:o + n].view_as(p).clone()}
        groups, k = [], 0
        for g in self.param_groups:
            d = {kk: vv for kk, vv in g.items() if kk != 'params'}
            d['params'] = list(range(k, k + len(g['params'])))
            k += len(g['params'])
            groups.append(d)
        return {'state': state if steps > 0 else {}, 'param_groups': groups}

    def load_state_dict(self, sd):
        for g, sg in zip(self.param_groups, sd['param_groups']):
            for kk, vv in sg.items():
                if kk != 'params':
                    g[kk] = vv
        for i, st in sd.get('state', {}).items():
            i = int(i)
            p, o = self.flat.params[i], self.flat.offsets[i]
            n = p.numel()
            self.exp_avg[o:o + n].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
            self.step_count = int(float(st['step']))
        self._norm[1].zero_()     # the loaded `step` counts effective steps: no skipped ones left to take off


def construct_optimizer(model, cfg):
    bn, non_bn = select_parameters(model, cfg)
    wd = cfg.OPTIMIZER.WEIGHT_DECAY
    groups = [{'params': bn, 'weight_decay': wd}, {'params': non_bn, 'weight_decay': wd}]
    lr = cfg.OPTIMIZER.LR.INITIAL_LR
    if cfg.OPTIMIZER.TYPE == 'AdamOptimizer':
        opt = FusedAdam(groups, lr=lr, betas=(0.9, 0.999), weight_decay=wd, fuse_groups=find_fuse_groups(model))
        mine = {id(p) for p in bn + non_bn}
        opt.packed_owners = [m for m in model.modules() if callable(getattr(m, 'invalidate_packed', None)) and
                             any(id(p) in mine for p in m.parameters())]
        return opt
    if cfg.OPTIMIZER.TYPE == 'MomentumOptimizer':
        return torch.optim.SGD(groups, lr=lr, momentum=0.9, weight_decay=wd)
    if cfg.OPTIMIZER.TYPE == 'AdamWOptimizer':
        return torch.optim.AdamW(groups, lr=lr, betas=(0.9, 0.999), weight_decay=wd)
    raise NotImplementedError('Does not support {} optimizer'.format(cfg.OPTIMIZER.TYPE))


def construct_scheduler(optimizer, cfg):
    """optimizer.py:79-104 (per-epoch schedulers)."""
    kind = cfg.OPTIMIZER.LR.DECAY_TYPE
    if kind == 'fixed':
        return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda epoch: 1)
    if kind == 'cosine':
        return torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=cfg.TRAIN.MAX_EPOCHS + 1, eta_min=0,
                                                          last_epoch=-1)
    if kind == 'cosinewarmup':
        base = cfg.OPTIMIZER.LR.INITIAL_LR
        nw = cfg.OPTIMIZER.LR.NUM_WARMUP_STEPS
        warm = np.linspace(cfg.OPTIMIZER.LR.WARMUP_LR / base, 1, nw)
        it = np.arange(cfg.TRAIN.MAX_EPOCHS + 1 - nw)
        fin = cfg.OPTIMIZER.LR.FINAL_LR / base
        cos = np.array([fin + 0.5 * (1 - fin) * (1 + math.cos(math.pi * t / len(it))) for t in it])
        sched = np.concatenate((warm, cos))
        return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda epoch: sched[epoch])
    if kind == 'multiply':
        dr = cfg.OPTIMIZER.LR.DECAY_RATE
        return torch.optim.lr_scheduler.MultiplicativeLR(optimizer, lr_lambda=lambda epoch: dr)
    raise NotImplementedError('Does not support {} scheduler'.format(kind))


def get_lr(optimizer):
    return [g['lr'] for g in optimizer.param_groups]


def set_lr(optimizer, new_lr):
    for g in optimizer.param_groups:
        g['lr'] = new_lr
