"""Functional ops of the MV-Former hot path: thin `torch.autograd.Function`s whose forward AND backward are
calls into libmvf_hip.so (include/mvf_hip.h).  PyTorch keeps the autograd graph, owns the memory (caching
allocator) and provides the stream; no arithmetic of the path is done by ATen here.

Every op fails loudly on non-CUDA tensors or a missing library (see _lib.py) -- there is no CPU fallback.
"""
import math
import os
import ctypes

import torch
import torch.distributed as dist

from . import _lib
from ._lib import call, ptr, stream, F32, BF16, FP8, F16


def _dt(dtype):
    if dtype in (torch.float32, 'fp32', 'f32', F32):
        return F32, torch.float32
    if dtype in (torch.bfloat16, 'bf16', BF16):
        return BF16, torch.bfloat16
    if dtype in ('fp8', 'mxfp8'):          # backbone only: MX-fp8 GEMM operands, bf16 activations / taps
        return FP8, torch.bfloat16
    if dtype in (torch.float16, 'fp16', 'f16', F16):   # frozen backbone only: IEEE fp16 operands / activations (the reference's
        return F16, torch.bfloat16                      # autocast dtype); the tapped features handed to the head stay bf16
    raise ValueError('unsupported compute dtype %r' % (dtype,))


def _mat(t):
    """2-D fp32 matrix with unit inner stride -> (data_ptr, row stride)."""
    if not t.is_cuda:
        raise _lib.MvfError('HIP op received a %s tensor (no CPU fallback)' % t.device)
    assert t.dim() == 2 and t.dtype == torch.float32, (t.shape, t.dtype)
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t, t.data_ptr(), t.stride(0) if t.shape[0] > 1 else t.shape[1]


def _hgemm(A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, bias=None, table=None, tab_si=0, tab_sn=0, tab_div=1,
           tab_mod=1, alpha=1.0, relu=False, accumulate=False):
    call('mvf_hgemm', A, sam, sak, B, sbk, sbn, C, ldc, ptr(bias), ptr(table), tab_si, tab_sn, tab_div, tab_mod,
         M, N, K, alpha, int(relu), int(accumulate), stream())


# ------------------------------------------------------------------------------------------------
# flat-gradient slots
# ------------------------------------------------------------------------------------------------
def grad_slot(p):
    """The parameter's slice of the flat gradient buffer (utils.distributed.FlatBuffers sets `p._mvf_grad`), or None.
    Backward kernels ACCUMULATE parameter gradients straight into it (the buffer is zeroed once per step) and the op
    returns None for that input: no per-parameter `grad.add_()` kernels, no temporaries."""
    if p is None:
        return None
    flat = getattr(p, '_mvf_flat', None)
    if flat is not None:
        flat.dirty = True         # a backward kernel is going to write here: the next zero_grad() has work to do
    return getattr(p, '_mvf_grad', None)


def grad_ready(*params):
    """Tells the gradient reducer (if any) that these parameters' slots are final -- what the post-accumulate-grad hook
    does for gradients that travel through autograd."""
    for p in params:
        if p is None:
            continue
        # runs at BACKWARD time, right after a kernel wrote the slot: forward -> zero_grad() -> backward -> (no step) ->
        # zero_grad() must find the buffer dirty (grad_slot() marked it at forward time only)
        flat = getattr(p, '_mvf_flat', None)
        if flat is not None:
            flat.dirty = True
        cb = getattr(p, '_mvf_ready', None)
        if cb is not None:
            cb(p)


# ------------------------------------------------------------------------------------------------
# Linear  y = [resid +] dropout(act(x W^T + b [+ table[(row // div) % mod]]))
# ------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    """fwd: one GEMM launch with bias / PE table / ReLU / dropout / residual in its epilogue.
    bwd: ONE launch (mvf_hlinear_bwd) for dX, dW and db (after one elementwise pass for a ReLU / dropout mask)."""

    @staticmethod
    def forward(ctx, x, w, b, relu, table, tab_div, tab_mod, resid, drop, slots, owners):
        x, xp, ldx = _mat(x)
        w, wp, ldw = _mat(w)
        M, K = x.shape
        N = w.shape[0]
        assert w.shape[1] == K
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        if table is not None:
            table = table.contiguous()
            assert table.shape[1] == N
        p, seed, off = drop if drop is not None else (0.0, 0, 0)
        assert not (relu and p > 0.0), 'ReLU and dropout in one epilogue are not needed on this path'
        rp, ldr = None, 0
        if resid is not None:
            resid, rp, ldr = _mat(resid)
        call('mvf_hgemm_ex', xp, ldx, 1, wp, 1, ldw, y.data_ptr(), N, ptr(b), ptr(table),
             N if table is not None else 0, 1, tab_div, tab_mod, M, N, K, 1.0, int(relu), 0, rp, ldr, float(p), seed, off,
             stream())
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.cfg = (relu, (float(p), seed, off), resid is not None, b is not None, slots, owners)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        relu, (p, seed, off), has_resid, has_bias, slots, owners = ctx.cfg
        M, K = x.shape
        N = w.shape[0]
        if dy.stride(1) != 1 or (dy.stride(0) != N and (relu or p > 0.0)) or dy.stride(0) < N:
            dy = dy.contiguous()      # the pre-op masks are indexed like the dense forward output; plain layers take dy with
        ldy = dy.stride(0) if M > 1 else N      # its row stride (a column slice of a wider gradient: concat_onehot's backward)
        dev = dy.device
        dx = torch.empty(M, K, device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        if slots is not None:
            gw, gb = slots
            acc, dw, db = 1, None, None
        else:
            dw = torch.empty(N, K, device=dev, dtype=torch.float32)
            db = torch.empty(N, device=dev, dtype=torch.float32) if has_bias else None
            gw, gb, acc = dw, db, 0
        d_resid = dy if has_resid else None
        if relu:                      # g = dy * [y > 0]
            gm = torch.empty_like(dy)
            call('mvf_relu_bwd', dy.data_ptr(), y.data_ptr(), gm.data_ptr(), M * N, stream())
            dy, ldy = gm, N
        elif p > 0.0:                 # g = dy * keep / (1 - p): the forward's counter-based mask
            gm = torch.empty_like(dy)
            call('mvf_dropout_add', dy.data_ptr(), None, gm.data_ptr(), M * N, p, seed, off, stream())
            dy, ldy = gm, N
        call('mvf_hlinear_bwd', dy.data_ptr(), ldy, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), ptr(dx), K,
             gw.data_ptr(), gw.stride(0) if gw.dim() == 2 else K, ptr(gb) if has_bias else None, M, N, K, acc, stream())
        if slots is not None:
            grad_ready(*owners)
        return dx, dw, db, None, None, None, None, d_resid, None, None, None


class _LinearTC(torch.autograd.Function):
    """y = x W^T + b [+ resid] for the big-M GEMMs of TRAINABLE backbone blocks in bf16 mode: forward and the input gradient
    run on the persistent bf16 MFMA kernel (operands cast to bf16 per call, fp32 accumulation, fp32 result through the
    read-modify epilogue); the weight / bias gradients stay in fp32 (mvf_hlinear_bwd without dX).  What fp16 autocast does
    to these layers in the reference, with an fp32 weight gradient.  Needs K % 128 == 0 and N % 128 == 0."""

    MC = 2048      # tokens per split-K chunk of the weight gradient

    @staticmethod
    def _bf16(t):
        o = torch.empty(t.shape, device=t.device, dtype=torch.bfloat16)
        call('mvf_cast_f32_bf16', ptr(t), ptr(o), t.numel(), stream())
        return o

    @staticmethod
    def _gemm(a_bf16, w_bf16, bias, out_f32):
        M, K = a_bf16.shape
        N = w_bf16.shape[0]
        call('mvf_gemm_tc', BF16, _lib.EPI_RESID, ptr(a_bf16), K, ptr(w_bf16), K, ptr(bias), None, 0, ptr(out_f32), N, None, 0,
             None, None, 1, M, N, K, stream())

    @staticmethod
    def forward(ctx, x, w, b, resid, slots, owners):
        x = x.contiguous()
        M, K = x.shape
        N = w.shape[0]
        y = resid.contiguous().clone() if resid is not None else torch.zeros(M, N, device=x.device, dtype=torch.float32)
        _LinearTC._gemm(_LinearTC._bf16(x), _LinearTC._bf16(w.detach().contiguous()), b, y)
        ctx.save_for_backward(x, w)
        ctx.cfg = (resid is not None, b is not None, slots, owners)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        has_resid, has_bias, slots, owners = ctx.cfg
        M, K = x.shape
        N = w.shape[0]
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.zeros(M, K, device=dy.device, dtype=torch.float32)
            _LinearTC._gemm(_LinearTC._bf16(dy), _LinearTC._bf16(w.detach().t().contiguous()), None, dx)   # dy . W
        if slots is not None:
            gw, gb = slots
            acc, dw, db = 1, None, None
        else:
            dw = torch.empty(N, K, device=dy.device, dtype=torch.float32)
            db = torch.empty(N, device=dy.device, dtype=torch.float32) if has_bias else None
            gw, gb, acc = dw, db, 0
        if N % 256 == 0 and K % 32 == 0 and gw.dim() == 2 and gw.stride(0) == K and M >= 4 * _LinearTC.MC:
            # dW = dY^T X on the bf16 kernel too, split-K over the token axis: both operands transposed (and cast) into
            # chunks of MC tokens, one batch of mvf_gemm_tc_batched per chunk with fp32 partial sums, summed in fixed order
            MC = _LinearTC.MC
            S = (M + MC - 1) // MC
            dyt = torch.empty(S * N, MC, device=dy.device, dtype=torch.bfloat16)
            xt = torch.empty(S * K, MC, device=dy.device, dtype=torch.bfloat16)
            call('mvf_transpose_chunks', ptr(dy), ptr(dyt), M, N, MC, stream())
            call('mvf_transpose_chunks', ptr(x), ptr(xt), M, K, MC, stream())
            part = torch.zeros(S * N, K, device=dy.device, dtype=torch.float32)
            call('mvf_gemm_tc_batched', _lib.EPI_RESID, ptr(dyt), MC, ptr(xt), MC, None, 0, ptr(part), K, S * N, K, MC, N, K,
                 stream())
            call('mvf_sum_batches', ptr(part), gw.data_ptr(), S, N * K, acc, stream())
            if has_bias:       # two-stage column sum (the one-stage kernel walks all M rows with N/64 workgroups: 5 ms)
                RS = 128
                bpart = torch.empty(RS, N, device=dy.device, dtype=torch.float32)
                call('mvf_colsum_split', ptr(dy), N, M, N, RS, ptr(bpart), stream())
                call('mvf_sum_batches', ptr(bpart), ptr(gb), RS, N, acc, stream())
        else:
            call('mvf_hlinear_bwd', dy.data_ptr(), N, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), None, K,
                 gw.data_ptr(), gw.stride(0) if gw.dim() == 2 else K, ptr(gb) if has_bias else None, M, N, K, acc, stream())
        if slots is not None:
            grad_ready(*owners)
        return dx, dw, db, (dy if has_resid else None), None, None


def linear_tc(x, w, b=None, resid=None):
    """ops.linear for the large GEMMs of trainable ViT blocks in bf16 mode (see _LinearTC)."""
    x2 = x.reshape(-1, x.shape[-1])
    slots, owners = None, ()
    if x2.requires_grad and grad_slot(w) is not None and (b is None or grad_slot(b) is not None):
        slots, owners = (grad_slot(w), grad_slot(b)), (w, b)
    r2 = None if resid is None else resid.reshape(-1, resid.shape[-1])
    return _LinearTC.apply(x2, w, b, r2, slots, owners).view(x.shape[:-1] + (w.shape[0],))


def linear(x, w, b=None, relu=False, table=None, tab_div=1, tab_mod=1, resid=None, drop=None, fused=None):
    """x [..., K] -> [..., N]; `table` [mod, N] is added to row r as table[(r // div) % mod] (sin/cos PE);
    `drop` = (p, seed, offset) applies the counter-based dropout to the result, `resid` [..., N] is added last.
    Parameter gradients go straight into their flat-gradient slots when the parameters have one (`fused` =
    (gw, gb, owners) supplies the slots of a weight/bias that are views over several parameters, e.g. Q|K|V)."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    slots, owners = None, ()
    if fused is not None:
        slots, owners = (fused[0], fused[1]), tuple(fused[2])
        for o in owners:
            grad_slot(o)                    # marks the flat gradient buffer as written
    elif x2.requires_grad and grad_slot(w) is not None and (b is None or grad_slot(b) is not None):
        slots, owners = (grad_slot(w), grad_slot(b)), (w, b)
    if slots is not None and not x2.requires_grad:
        slots, owners = None, ()            # backward would never run: let autograd handle the parameters
    r2 = resid.reshape(-1, w.shape[0]) if resid is not None else None
    y = _Linear.apply(x2, w, b, relu, table, tab_div, tab_mod, r2, drop, slots, owners)
    return y.view(*lead, w.shape[0])


class _MatmulNN(torch.autograd.Function):
    """c = a @ b for small fp32 matrices (wq = q W_K of the pooling rewrite)."""

    @staticmethod
    def forward(ctx, a, b):
        a, ap, lda = _mat(a)
        b, bp, ldb = _mat(b)
        M, K = a.shape
        N = b.shape[1]
        c = torch.empty(M, N, device=a.device, dtype=torch.float32)
        _hgemm(ap, lda, 1, bp, ldb, 1, c.data_ptr(), N, M, N, K)
        ctx.save_for_backward(a, b)
        return c

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        dc, dcp, ldc = _mat(dc)
        M, K = a.shape
        N = b.shape[1]
        da = db = None
        if ctx.needs_input_grad[0]:
            da = torch.empty(M, K, device=dc.device, dtype=torch.float32)
            _hgemm(dcp, ldc, 1, b.data_ptr(), 1, b.stride(0), da.data_ptr(), K, M, K, N)     # da = dc . b^T
        if ctx.needs_input_grad[1]:
            db = torch.empty(K, N, device=dc.device, dtype=torch.float32)
            _hgemm(a.data_ptr(), 1, a.stride(0), dcp, ldc, 1, db.data_ptr(), N, K, N, M)     # db = a^T . dc
        return da, db


def matmul(a, b):
    return _MatmulNN.apply(a, b)


class _StaticQuery(torch.autograd.Function):
    """wq = (Q_s + Q_s_b) W_K  [nq, C]: the query side of the pooling rewrite for static queries (LSTPCrossAtt,
    mvformer.py:383 `Q = self.Q_s + self.Q_s_b`, folded through linear_K2d as csrc/lstp_pool.hip describes).
    One launch each way (csrc/lstp_pool.hip: mvf_static_query_fwd / _bwd); the backward accumulates all three parameter
    gradients straight into their flat-gradient slots -- no broadcast add, no select / sum / AccumulateGrad kernels of the
    autograd engine."""

    @staticmethod
    def forward(ctx, qs, qb, wk, slots):
        ctx.owners = (qs, qb, wk)
        nq, d = qs.shape[-2], qs.shape[-1]
        C = wk.shape[1]
        dev = wk.device
        out = torch.empty(nq, C, device=dev, dtype=torch.float32)
        call('mvf_static_query_fwd', qs.data_ptr(), qb.data_ptr(), wk.data_ptr(), wk.stride(0), out.data_ptr(), nq, d, C, stream())
        ctx.save_for_backward(qs, qb, wk)
        ctx.slots = slots
        return out

    @staticmethod
    def backward(ctx, dv):
        qs, qb, wk = ctx.saved_tensors
        gqs, gqb, gwk = ctx.slots
        nq, d = qs.shape[-2], qs.shape[-1]
        C = wk.shape[1]
        dv, dvp, ldv = _mat(dv)
        call('mvf_static_query_bwd', dvp, ldv, qs.data_ptr(), qb.data_ptr(), wk.data_ptr(), wk.stride(0), gqs.data_ptr(), gqb.data_ptr(),
             gwk.data_ptr(), gwk.stride(0), nq, d, C, stream())
        grad_ready(*ctx.owners)
        return None, None, None, None


def static_query(q_s, q_s_b, w_k):
    """(Q_s + Q_s_b)[0] @ W_K for Q_s [1, nq, d], Q_s_b [d], W_K [d, C]: fused form with in-slot gradients when all three
    parameters live in the flat buffers (training under FusedAdam), the plain composition otherwise."""
    slots = (grad_slot(q_s), grad_slot(q_s_b), grad_slot(w_k))
    if any(s is None for s in slots) or not (q_s.requires_grad and q_s_b.requires_grad and w_k.requires_grad) or \
            not (q_s.is_contiguous() and q_s_b.is_contiguous() and w_k.stride(1) == 1) or q_s.shape[-2] > 8:
        return matmul((q_s + q_s_b)[0], w_k)
    out = _StaticQuery.apply(q_s, q_s_b, w_k, slots)
    return out


# ------------------------------------------------------------------------------------------------
# LayerNorm
# ------------------------------------------------------------------------------------------------
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, eps):
        x = x.contiguous()
        R, D = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(R, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        call('mvf_ln_fwd', ptr(x), ptr(g), ptr(b), ptr(y), ptr(mean), ptr(rstd), R, D, eps, stream())
        ctx.save_for_backward(x, g, mean, rstd)
        use = x.requires_grad and grad_slot(g) is not None and grad_slot(b) is not None
        ctx.slots = (grad_slot(g), grad_slot(b)) if use else (None, None)
        ctx.owners = (g, b) if use else ()
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        R, D = x.shape
        dx = torch.empty_like(x)
        gw, gb = ctx.slots
        if gw is not None:          # accumulate dgamma/dbeta into the flat gradient buffer
            call('mvf_ln_bwd', ptr(dy), ptr(x), ptr(g), ptr(mean), ptr(rstd), ptr(dx), gw.data_ptr(), gb.data_ptr(), R, D,
                 0, 1, stream())
            grad_ready(*ctx.owners)
            return dx, None, None, None
        dg = torch.empty_like(g)
        db = torch.empty_like(g)
        call('mvf_ln_bwd', ptr(dy), ptr(x), ptr(g), ptr(mean), ptr(rstd), ptr(dx), ptr(dg), ptr(db), R, D, 0, 0, stream())
        return dx, dg, db, None


def layer_norm(x, g, b, eps=1e-5):
    shp = x.shape
    return _LayerNorm.apply(x.reshape(-1, shp[-1]), g, b, eps).view(shp)


class _LayerNormFork(torch.autograd.Function):
    """(x, LN(x)) for a pre-LN residual connection out = x + sub(LN(x)) (ResidualConnection, models/utils.py:147-159).
    x has two consumers there -- the LayerNorm and the residual add -- and the autograd engine would sum their two gradients
    with an elementwise ATen kernel.  Here x leaves through this node as well, so BOTH gradients come back to it and the
    LayerNorm backward kernel adds the residual one in its own pass (mvf_ln_bwd_res)."""

    @staticmethod
    def forward(ctx, x, g, b, eps):
        x = x.contiguous()
        R, D = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(R, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        call('mvf_ln_fwd', ptr(x), ptr(g), ptr(b), ptr(y), ptr(mean), ptr(rstd), R, D, eps, stream())
        ctx.save_for_backward(x, g, mean, rstd)
        use = x.requires_grad and grad_slot(g) is not None and grad_slot(b) is not None
        ctx.slots = (grad_slot(g), grad_slot(b)) if use else (None, None)
        ctx.owners = (g, b) if use else ()
        ctx.set_materialize_grads(False)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, dres, dy):
        x, g, mean, rstd = ctx.saved_tensors
        if dy is None:                     # the LayerNorm branch was not used
            return dres, None, None, None
        dy = dy.contiguous()
        R, D = x.shape
        dx = torch.empty_like(x)
        gw, gb = ctx.slots
        acc = 1 if gw is not None else 0
        dg = gw if gw is not None else torch.empty_like(g)
        db = gb if gb is not None else torch.empty_like(g)
        if dres is not None:
            dres = dres.contiguous()
            call('mvf_ln_bwd_res', ptr(dy), ptr(x), ptr(g), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), dg.data_ptr(), db.data_ptr(),
                 R, D, acc, stream())
        else:
            call('mvf_ln_bwd', ptr(dy), ptr(x), ptr(g), ptr(mean), ptr(rstd), ptr(dx), dg.data_ptr(), db.data_ptr(), R, D, 0, acc,
                 stream())
        if gw is not None:
            grad_ready(*ctx.owners)
            return dx, None, None, None
        return dx, dg, db, None


def layer_norm_fork(x, g, b, eps=1e-5):
    """-> (x, LN(x)): use the returned x for the residual add (see _LayerNormFork)."""
    shp = x.shape
    xr, y = _LayerNormFork.apply(x.reshape(-1, shp[-1]), g, b, eps)
    return xr.view(shp), y.view(shp)


# ------------------------------------------------------------------------------------------------
# exact-erf GELU (trainable ViT blocks)
# ------------------------------------------------------------------------------------------------
class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        call('mvf_gelu_fwd', ptr(x), ptr(y), x.numel(), stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        call('mvf_gelu_bwd', ptr(dy), ptr(x), ptr(dx), x.numel(), stream())
        return dx


def gelu(x):
    return _Gelu.apply(x)


class _LayerScaleAdd(torch.autograd.Function):
    """resid + y * gamma (timm LayerScale + the block's residual add), rows x D fp32."""

    @staticmethod
    def forward(ctx, y, gamma, resid):
        y, resid = y.contiguous(), resid.contiguous()
        R, D = y.shape
        out = torch.empty_like(y)
        call('mvf_colscale', ptr(y), ptr(gamma), ptr(resid), ptr(out), R, D, 0, stream())
        ctx.save_for_backward(y, gamma)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, gamma = ctx.saved_tensors
        dout = dout.contiguous()
        R, D = y.shape
        dy = torch.empty_like(y)
        call('mvf_colscale', ptr(dout), ptr(gamma), None, ptr(dy), R, D, 1, stream())
        prod = torch.empty_like(y)
        call('mvf_colscale', ptr(dout), None, ptr(y), ptr(prod), R, D, 2, stream())
        dgamma = torch.empty_like(gamma)
        call('mvf_colsum', ptr(prod), D, R, D, ptr(dgamma), 0, stream())
        return dy, dgamma, dout


def layerscale_add(y, gamma, resid):
    return _LayerScaleAdd.apply(y, gamma, resid)


# ------------------------------------------------------------------------------------------------
# BatchNorm1d (+ optional fused ReLU), with cross-rank statistics when `group_size > 1` (SyncBN)
# ------------------------------------------------------------------------------------------------
def _sync_world(group):
    """Ranks a SyncBatchNorm exchanges statistics with; 0 = no exchange (no process group, or one rank and not forced)."""
    from .utils.distributed import collectives_active
    return dist.get_world_size(group) if collectives_active() else 0


_SYNCBN_BUFS = {}


def syncbn_local_buffer(C, count, dev):
    """The block a rank contributes to SyncBatchNorm's exchange, [mean C | biased var C | row count]: one persistent buffer per
    (C, count, device) whose last word is written ONCE -- the statistics kernels write mean / var straight into its first 2C words
    (views `[:C]`, `[C:2C]`), so the step needs no cat / fill kernel to assemble it."""
    key = (int(C), float(count), str(dev))
    b = _SYNCBN_BUFS.get(key)
    if b is None:
        b = torch.zeros(2 * C + 1, device=dev, dtype=torch.float32)
        b[2 * C] = float(count)
        _SYNCBN_BUFS[key] = b
    return b


def sync_bn_stats(mean, var, count, group=None, equal_counts=False, running=None, local=None):
    """Cross-rank batch statistics for SyncBatchNorm (train.py:283): every rank contributes (mean, biased var, row
    count) of its local rows -- 2C+1 floats, collective C2 of SURVEY.md -- and all ranks merge them with Chan's
    parallel-variance formula, which equals BatchNorm statistics over the rank-concatenated batch.
    `equal_counts`: every rank holds `count` rows (the data-parallel step: B x V x T x entities rows per rank, fixed by the
    config), so the global count is count x world ON THE HOST -- no device-to-host read, which would stall the host until
    the side-stream backbone forward the head waits for has finished and so undo the one-batch lookahead on every BatchNorm
    of every step.  Ragged counts (equal_counts=False) read the gathered total back (one sync).
    running = (running_mean, running_var, momentum) or None: updated from the merged statistics like nn.BatchNorm1d in training.
    On the device with equal counts (the training step): ONE all-gather into a [W, 2C+1] block + ONE kernel (mvf_syncbn_merge:
    merge + running update) -- `local` is the rank's block from syncbn_local_buffer() when the caller had its statistics kernel write
    there (else it is assembled here).  On CPU tensors (gloo tests, --plumbing) and for ragged counts: plain torch."""
    C = mean.numel()
    world = dist.get_world_size(group)
    if mean.is_cuda and equal_counts:
        if local is None:
            local = syncbn_local_buffer(C, count, mean.device)
            local[:C].copy_(mean)
            local[C:2 * C].copy_(var)
        gathered = torch.empty(world * (2 * C + 1), device=mean.device, dtype=torch.float32)   # flat: gloo's form of the call takes no 2-D output
        dist.all_gather_into_tensor(gathered, local, group=group)
        gm = torch.empty(C, device=mean.device, dtype=torch.float32)
        gv = torch.empty_like(gm)
        rm, rv, mom = running if running is not None else (None, None, 0.0)
        call('mvf_syncbn_merge', ptr(gathered), world, C, float(count), ptr(gm), ptr(gv), ptr(rm), ptr(rv), float(mom), stream())
        return gm, gv, float(count) * world
    local = torch.cat([mean, var, mean.new_tensor([float(count)])])
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local, group=group)
    st = torch.stack(gathered)
    n = st[:, -1:]
    if equal_counts:
        total = float(count) * world
        if os.environ.get('MVF_CHECK_SYNCBN', '0') == '1':
            assert float(n.sum()) == total, 'SyncBN: ranks hold different row counts'
        # equal weights: plain means over the ranks (exactly the local statistics when world == 1)
        gm = st[:, :C].sum(0) / world
        gv = (st[:, C:2 * C] + (st[:, :C] - gm) ** 2).sum(0) / world
    else:
        total = float(n.sum())
        gm = (st[:, :C] * n).sum(0) / total
        gv = ((st[:, C:2 * C] + (st[:, :C] - gm) ** 2) * n).sum(0) / total
    if running is not None and running[0] is not None:
        rm, rv, mom = running
        with torch.no_grad():
            rm.mul_(1 - mom).add_(gm, alpha=mom)
            rv.mul_(1 - mom).add_(gv, alpha=mom * total / max(total - 1.0, 1.0))
    return gm.contiguous(), gv.contiguous(), total


def _bn_ws(R, C, dev):
    return torch.empty(_lib.load().mvf_bn_workspace_floats(R, C), device=dev, dtype=torch.float32)


class _BatchNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, running_mean, running_var, momentum, eps, relu, sync, group):
        x = x.contiguous()
        R, C = x.shape
        count = float(R)
        world = _sync_world(group) if sync else 0
        local = None
        if world > 0:                    # SyncBatchNorm: the statistics land in this rank's block of the exchange
            local = syncbn_local_buffer(C, count, x.device)
            mean, var = local[:C], local[C:2 * C]
        else:
            mean = torch.empty(C, device=x.device, dtype=torch.float32)
            var = torch.empty_like(mean)
        ws = _bn_ws(R, C, x.device)
        local_running = world == 0       # the statistics kernel updates the running buffers itself
        call('mvf_bn_stats', ptr(x), R, C, ptr(mean), ptr(var), ptr(running_mean) if local_running else None,
             ptr(running_var) if local_running else None, float(momentum), ptr(ws), ws.numel(), stream())
        if world > 0:
            # exchange (mean, biased var, count) and merge (Chan); 2C+1 floats per rank -- collective C2 of SURVEY.md; the merge kernel
            # also moves the running buffers
            mean, var, count = sync_bn_stats(mean, var, count, group, equal_counts=True,
                                             running=(running_mean, running_var, momentum), local=local)
        y = torch.empty_like(x)
        call('mvf_bn_fwd', ptr(x), ptr(mean), ptr(var), ptr(g), ptr(b), ptr(y), R, C, eps, int(relu), stream())
        ctx.save_for_backward(x, g, b, mean, var)
        use = x.requires_grad and grad_slot(g) is not None and grad_slot(b) is not None
        ctx.cfg = (eps, relu, count, world, group, (grad_slot(g), grad_slot(b)) if use else None, (g, b) if use else ())
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, b, mean, var = ctx.saved_tensors
        eps, relu, count, world, group, slots, owners = ctx.cfg
        dy = dy.contiguous()
        R, C = x.shape
        s = torch.empty(2, C, device=x.device, dtype=torch.float32)
        ws = _bn_ws(R, C, x.device)
        if slots is not None:            # local dgamma / dbeta straight into the flat gradient buffer
            dgamma = dbeta = None
            gp, bp, acc = slots[0].data_ptr(), slots[1].data_ptr(), 1
        else:
            dgamma, dbeta = torch.empty_like(g), torch.empty_like(b)
            gp, bp, acc = dgamma.data_ptr(), dbeta.data_ptr(), 0
        call('mvf_bn_bwd_reduce', ptr(dy), ptr(x), ptr(mean), ptr(var), ptr(g), ptr(b), s[0].data_ptr(), s[1].data_ptr(),
             gp, bp, acc, R, C, eps, int(relu), ptr(ws), ws.numel(), stream())
        if slots is not None:
            grad_ready(*owners)
        if world > 0:
            dist.all_reduce(s, group=group)          # collective C3 of SURVEY.md
        dx = torch.empty_like(x)
        call('mvf_bn_bwd_apply', ptr(dy), ptr(x), ptr(mean), ptr(var), ptr(g), ptr(b), s[0].data_ptr(),
             s[1].data_ptr(), ptr(dx), R, C, eps, int(relu), count, stream())
        return dx, dgamma, dbeta, None, None, None, None, None, None, None


class _BatchNormEval(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, mean, var, eps, relu):
        x = x.contiguous()
        R, C = x.shape
        y = torch.empty_like(x)
        call('mvf_bn_fwd', ptr(x), ptr(mean), ptr(var), ptr(g), ptr(b), ptr(y), R, C, eps, int(relu), stream())
        ctx.save_for_backward(x, g, b, mean, var)
        ctx.cfg = (eps, relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, b, mean, var = ctx.saved_tensors
        eps, relu = ctx.cfg
        dy = dy.contiguous()
        R, C = x.shape
        s = torch.empty(2, C, device=x.device, dtype=torch.float32)
        ws = _bn_ws(R, C, x.device)
        call('mvf_bn_bwd_reduce', ptr(dy), ptr(x), ptr(mean), ptr(var), ptr(g), ptr(b), s[0].data_ptr(), s[1].data_ptr(),
             None, None, 0, R, C, eps, int(relu), ptr(ws), ws.numel(), stream())
        dx = torch.empty_like(x)
        call('mvf_bn_bwd_apply', ptr(dy), ptr(x), ptr(mean), ptr(var), ptr(g), ptr(b), s[0].data_ptr(),
             s[1].data_ptr(), ptr(dx), R, C, eps, int(relu), 0.0, stream())
        return dx, s[1], s[0], None, None, None, None


def batch_norm(x, g, b, running_mean, running_var, training, momentum=0.1, eps=1e-5, relu=False, sync=False,
               group=None):
    if training:
        return _BatchNormTrain.apply(x, g, b, running_mean, running_var, momentum, eps, relu, sync, group)
    return _BatchNormEval.apply(x, g, b, running_mean, running_var, eps, relu)


# ------------------------------------------------------------------------------------------------
# Temporal multi-head self-attention core
# ------------------------------------------------------------------------------------------------
class _TemporalAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, mask, B, S, H):
        qkv = qkv.contiguous()
        Dm = qkv.shape[1] // 3
        o = torch.empty(B * S, Dm, device=qkv.device, dtype=torch.float32)
        lse = torch.empty(B, H, S, device=qkv.device, dtype=torch.float32)
        mlen = S
        if mask is not None:
            # [B, S] key mask, or [B, S / k]: the frame mask of a joint (entity, frame) sequence, read periodically by the kernels
            # (no tiled copy).  No kernel runs here when the mask already is contiguous fp32.
            mask = mask.reshape(B, -1)
            mlen = mask.shape[1]
            assert S % mlen == 0, (S, mlen)
            if mask.dtype != torch.float32 or not mask.is_contiguous():
                mask = mask.contiguous().float()
        call('mvf_tattn_fwd', ptr(qkv), ptr(mask), mlen, ptr(o), ptr(lse), B, S, H, Dm, stream())
        ctx.save_for_backward(qkv, mask, o, lse)
        ctx.dims = (B, S, H, Dm, mlen)
        return o

    @staticmethod
    def backward(ctx, d_o):
        qkv, mask, o, lse = ctx.saved_tensors
        B, S, H, Dm, mlen = ctx.dims
        d_o = d_o.contiguous()
        dqkv = torch.empty_like(qkv)
        call('mvf_tattn_bwd', ptr(qkv), ptr(mask), mlen, ptr(o), ptr(lse), ptr(d_o), ptr(dqkv), B, S, H, Dm, stream())
        return dqkv, None, None, None, None


class _VitAttentionBF16(torch.autograd.Function):
    """Self-attention core of a TRAINABLE ViT block in bf16 mode (no mask, head dim 64): bf16 flash forward that keeps the
    per-query log-sum-exp, bf16 MFMA backward (csrc/vit_attn_bwd.hip).  fp32 in / out, so it drops in where
    temporal_attention (fp32 matrix cores, 1/16 of the rate) is used in parity mode."""

    @staticmethod
    def forward(ctx, qkv, F, N, H):
        qkv = qkv.contiguous()
        D = qkv.shape[1] // 3
        dev = qkv.device
        q16 = torch.empty(qkv.shape, device=dev, dtype=torch.bfloat16)
        call('mvf_cast_f32_bf16', ptr(qkv), ptr(q16), qkv.numel(), stream())
        o16 = torch.empty(F * N, D, device=dev, dtype=torch.bfloat16)
        npad = (N + 15) // 16 * 16
        lse = torch.zeros(F, H, npad, device=dev, dtype=torch.float32)
        call('mvf_vit_attn_fwd_lse', ptr(q16), ptr(o16), ptr(lse), F, N, H, D, stream())
        o = torch.empty(F * N, D, device=dev, dtype=torch.float32)
        call('mvf_cast_bf16_f32', ptr(o16), ptr(o), o.numel(), stream())
        ctx.save_for_backward(q16, o16, lse)
        ctx.dims = (F, N, H, D)
        return o

    @staticmethod
    def backward(ctx, d_o):
        q16, o16, lse = ctx.saved_tensors
        F, N, H, D = ctx.dims
        d_o = d_o.contiguous()
        d16 = torch.empty(d_o.shape, device=d_o.device, dtype=torch.bfloat16)
        call('mvf_cast_f32_bf16', ptr(d_o), ptr(d16), d_o.numel(), stream())
        delta = torch.empty_like(lse)
        dqkv = torch.empty(F * N, 3 * D, device=d_o.device, dtype=torch.float32)
        call('mvf_vit_attn_bwd', ptr(q16), ptr(o16), ptr(d16), ptr(lse), ptr(delta), ptr(dqkv), F32, F, N, H, D, stream())
        return dqkv, None, None, None


def vit_attention_bf16(qkv, F, N, H):
    """qkv [F*N, 3*D] fp32 (q | k | v column blocks, heads of 64 contiguous inside) -> [F*N, D] fp32."""
    return _VitAttentionBF16.apply(qkv, F, N, H)


class _ViTBlockTC(torch.autograd.Function):
    """One TRAINABLE timm Block (x + proj(attn(norm1 x)); x + fc2(gelu(fc1(norm2 x)))) in bf16 mode as ONE autograd node
    (ViTBackEnd, reference models/transformer.py:364-392, under fp16 autocast there).  The residual stream stays fp32; every
    GEMM operand is bf16 and is written once by its producer (LayerNorm, the previous GEMM's epilogue, GELU, attention) --
    no per-GEMM casts, no zero-filled outputs, no fp32 copies of the wide activations:
      forward   4 GEMMs on the persistent bf16 kernel (qkv / fc1: bf16 result; proj / fc2: fp32 = residual + result, out of place)
      backward  per linear layer: dX = dY W (same kernel, W^T cast once), dW = dY^T X split-K over chunks of MC tokens
                (mvf_grad_prep writes the token-major operands and the bias-gradient partial sums in one pass),
                bf16 flash-attention backward, LayerNorm backward with the residual gradient added in the same pass.
    Parameter gradients are fp32 and go straight into the flat-gradient slots when every parameter has one."""

    MC = int(os.environ.get('MVF_BLOCK_MC', '2048'))      # tokens per split-K chunk of the weight gradients

    @staticmethod
    def _bf(dev, *shape):
        return torch.empty(*shape, device=dev, dtype=torch.bfloat16)

    @staticmethod
    def _store(a, w, bias, M, N, K):
        """bf16 [M, N] = a [M, K] w[N, K]^T + bias"""
        c = _ViTBlockTC._bf(a.device, M, N)
        call('mvf_gemm_tc', BF16, _lib.EPI_STORE, ptr(a), K, ptr(w), K, ptr(bias), ptr(c), N, None, 0, None, 0, None, None, 1,
             M, N, K, stream())
        return c

    @staticmethod
    def _f32(a, w, bias, addend, M, N, K):
        """fp32 [M, N] = a w^T + bias [+ addend]"""
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
        call('mvf_gemm_tc_f32', ptr(a), K, ptr(w), K, ptr(bias), ptr(out), N, ptr(addend), M, N, K, stream())
        return out

    @staticmethod
    def _wt(w):
        """fp32 [N, K] -> bf16 [K, N] (the operand of dX = dY W)"""
        N, K = w.shape
        o = _ViTBlockTC._bf(w.device, K, N)
        call('mvf_transpose_chunks', ptr(w), ptr(o), N, K, N, stream())
        return o

    @staticmethod
    def forward(ctx, x, heads, eps1, eps2, *params):
        n1w, n1b, qkvw, qkvb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b = params
        F_, N_, D = x.shape
        M, Hd = F_ * N_, f1w.shape[0]
        x = x.contiguous().view(M, D)
        dev, T = x.device, _ViTBlockTC
        wq, wp, w1, w2 = (_LinearTC._bf16(w.detach().contiguous()) for w in (qkvw, pw, f1w, f2w))
        h1 = T._bf(dev, M, D)
        call('mvf_layernorm_fwd', BF16, ptr(x), D, ptr(n1w), ptr(n1b), ptr(h1), D, M, D, eps1, stream())
        qkv = T._store(h1, wq, qkvb, M, 3 * D, D)
        o = T._bf(dev, M, D)
        lse = torch.empty(F_, heads, (N_ + 15) // 16 * 16, device=dev, dtype=torch.float32)
        call('mvf_vit_attn_fwd_lse', ptr(qkv), ptr(o), ptr(lse), F_, N_, heads, D, stream())
        x1 = T._f32(o, wp, pb, x, M, D, D)
        h2 = T._bf(dev, M, D)
        call('mvf_layernorm_fwd', BF16, ptr(x1), D, ptr(n2w), ptr(n2b), ptr(h2), D, M, D, eps2, stream())
        u = T._store(h2, w1, f1b, M, Hd, D)
        g = T._bf(dev, M, Hd)
        call('mvf_gelu_bf16', ptr(u), ptr(g), u.numel(), stream())
        y = T._f32(g, w2, f2b, x1, M, D, Hd)
        ctx.save_for_backward(x, x1, h1, qkv, o, lse, h2, u, g, *params)
        ctx.cfg = (F_, N_, heads, eps1, eps2)
        ctx.use_slots = all(grad_slot(p) is not None for p in params)
        return y.view(F_, N_, D)

    @staticmethod
    def _linear_bwd(ctx, dy_t, dy_bpart, xin, w, b, grads, iw, ib, M):
        """dW (+)= dy^T x from the token-major chunks `dy_t` [S*N, MC] and x's (made here), db from the column-sum partials"""
        T, MC = _ViTBlockTC, _ViTBlockTC.MC
        N, K = w.shape
        S = (M + MC - 1) // MC
        xt = T._bf(w.device, S * K, MC)
        call('mvf_grad_prep', BF16, ptr(xin), None, ptr(xt), None, 0, M, K, MC, stream())
        part = torch.empty(S * N, K, device=w.device, dtype=torch.float32)
        call('mvf_gemm_tc_batched_f32', ptr(dy_t), MC, ptr(xt), MC, ptr(part), K, S * N, K, MC, N, K, stream())
        for t, src, n, idx, owner in ((part, S, N * K, iw, w), (dy_bpart, dy_bpart.shape[0], N, ib, b)):
            if owner is None:
                continue
            if ctx.use_slots:
                call('mvf_sum_batches', ptr(t), grad_slot(owner).data_ptr(), src, n, 1, stream())
                grad_ready(owner)
            else:
                grads[idx] = torch.empty_like(owner)
                call('mvf_sum_batches', ptr(t), ptr(grads[idx]), src, n, 0, stream())

    @staticmethod
    def backward(ctx, dy):
        x, x1, h1, qkv, o, lse, h2, u, g = ctx.saved_tensors[:9]
        params = ctx.saved_tensors[9:]
        n1w, n1b, qkvw, qkvb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b = params
        F_, N_, heads, eps1, eps2 = ctx.cfg
        M, D = x.shape
        Hd = f1w.shape[0]
        dev, T, MC = x.device, _ViTBlockTC, _ViTBlockTC.MC
        S = (M + MC - 1) // MC
        PR = S * MC // 256                   # partial column sums: one row per 256 (zero-padded) tokens
        grads = [None] * 12
        dy = dy.contiguous().view(M, D)

        def prep(t, dt, C, rowmajor):
            tt = T._bf(dev, S * C, MC)
            part = torch.empty(PR, C, device=dev, dtype=torch.float32)
            rm = T._bf(dev, M, C) if rowmajor else None
            call('mvf_grad_prep', dt, ptr(t), ptr(rm), ptr(tt), ptr(part), PR, M, C, MC, stream())
            return rm, tt, part

        def ln_bwd(dh, xin, gam, dres, want_bf16, ig, ib, eps):
            dx = torch.empty(M, D, device=dev, dtype=torch.float32)
            dxb = T._bf(dev, M, D) if want_bf16 else None
            if ctx.use_slots:
                gg, gb = grad_slot(params[ig]), grad_slot(params[ib])
            else:
                gg = grads[ig] = torch.zeros_like(gam)
                gb = grads[ib] = torch.zeros_like(gam)
            call('mvf_ln_bwd_block', ptr(dh), ptr(xin), ptr(gam), ptr(dres), ptr(dx), ptr(dxb), gg.data_ptr(), gb.data_ptr(),
                 M, D, eps, stream())
            if ctx.use_slots:
                grad_ready(params[ig], params[ib])
            return dx, dxb

        # fc2: y = x1 + g W2^T + b2
        dyb, dyt, dyp = prep(dy, F32, D, True)
        dg = T._store(dyb, T._wt(f2w), None, M, Hd, D)
        T._linear_bwd(ctx, dyt, dyp, g, f2w, f2b, grads, 10, 11, M)
        du = T._bf(dev, M, Hd)
        call('mvf_gelu_bwd_bf16', ptr(dg), ptr(u), ptr(du), du.numel(), stream())
        del dg
        # fc1: u = h2 W1^T + b1
        _, dut, dup = prep(du, BF16, Hd, False)
        dh2 = T._f32(du, T._wt(f1w), None, None, M, D, Hd)
        T._linear_bwd(ctx, dut, dup, h2, f1w, f1b, grads, 8, 9, M)
        del du, dut
        # norm2 + the residual branch: dx1 = dy + d LN2
        dx1, dx1b = ln_bwd(dh2, x1, n2w, dy, True, 6, 7, eps2)
        # proj: x1 = x + o Wp^T + bp
        _, dx1t, dx1p = prep(dx1b, BF16, D, False)
        d_o = T._store(dx1b, T._wt(pw), None, M, D, D)
        T._linear_bwd(ctx, dx1t, dx1p, o, pw, pb, grads, 4, 5, M)
        # attention core
        delta = torch.empty_like(lse)
        dqkv = T._bf(dev, M, 3 * D)
        call('mvf_vit_attn_bwd', ptr(qkv), ptr(o), ptr(d_o), ptr(lse), ptr(delta), ptr(dqkv), BF16, F_, N_, heads, D, stream())
        # qkv: qkv = h1 Wq^T + bq
        _, dqt, dqp = prep(dqkv, BF16, 3 * D, False)
        dh1 = T._f32(dqkv, T._wt(qkvw), None, None, M, D, 3 * D)
        T._linear_bwd(ctx, dqt, dqp, h1, qkvw, qkvb, grads, 2, 3, M)
        # norm1 + the residual branch
        dx, _ = ln_bwd(dh1, x, n1w, dx1, False, 0, 1, eps1)
        return (dx.view(F_, N_, D), None, None, None) + tuple(grads)


def vit_block_tc_supported(D, heads, hidden):
    """Shapes the fused trainable block takes: head dim 64, every GEMM dimension a multiple of the 256-row batch of the split-K
    weight gradient."""
    return D == heads * 64 and D % 256 == 0 and hidden % 256 == 0


def vit_block_tc(x, heads, eps1, eps2, params):
    """x [F, N, D] fp32 -> [F, N, D] fp32; params = (norm1.weight, norm1.bias, qkv.weight, qkv.bias, proj.weight, proj.bias,
    norm2.weight, norm2.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias) of a timm Block without LayerScale."""
    return _ViTBlockTC.apply(x, heads, eps1, eps2, *params)


def temporal_attention(qkv, mask, B, S, H):
    """qkv [B*S, 3*Dm] (q | k | v column blocks, heads contiguous inside) -> [B*S, Dm]; mask [B, S] (0 = masked key) or
    [B, S / k] (then key s is masked by column s % (S / k): the frame mask of k entities' joint sequence)."""
    return _TemporalAttention.apply(qkv, mask, B, S, H)


# ------------------------------------------------------------------------------------------------
# Row-chain kernels of the head on the 16-bit matrix cores (csrc/head_chain.hip)
# ------------------------------------------------------------------------------------------------
# MVF_HEAD_CHAIN=0: keep the one-kernel-per-operator fp32 head everywhere (A/B measurements, tests)
HEAD_CHAIN = os.environ.get('MVF_HEAD_CHAIN', '1') != '0'


def head_dtype_of(cfg):
    """'fp16' | 'bf16' | 'fp32': the trainable head's GEMM operand dtype.  cfg.MI355X.HEAD_DTYPE when given; else fp16 when the backbone
    runs in a reduced precision (bf16, MX-fp8, fp16, or USE_AMP without a COMPUTE_DTYPE) -- the reference's head runs under fp16
    autocast then (train.py:113-117) -- and fp32 in parity mode (COMPUTE_DTYPE fp32).
    'fp16' (round 6): the row-chain kernels with IEEE fp16 operands in the FORWARD GEMMs and bf16 operands in every gradient GEMM (no
    loss scaler needed: csrc/head_chain.hip).  Measured at configs[1] size (tools/fp16_error_budget.py, profiles/r06): per-frame
    embeddings 5e-4 of the fp32 oracle whatever the backbone's dtype, against 4e-3 with bf16 forward operands, at the same speed.
    'bf16': every head GEMM on bf16 operands (rounds 5's head)."""
    mi = cfg.MI355X if 'MI355X' in cfg else {}
    if 'HEAD_DTYPE' in mi:
        hd = str(mi['HEAD_DTYPE']).lower()
        if hd not in ('bf16', 'fp16', 'fp32'):
            raise ValueError("MI355X.HEAD_DTYPE must be 'bf16', 'fp16' or 'fp32' (got %r)" % (mi['HEAD_DTYPE'],))
        return hd
    cd = str(mi['COMPUTE_DTYPE'] if 'COMPUTE_DTYPE' in mi else ('bf16' if ('USE_AMP' in cfg and cfg.USE_AMP) else 'fp32')).lower()
    return 'fp16' if cd in ('bf16', 'fp8', 'mxfp8', 'fp16', 'f16') else 'fp32'


def chain_dtype(hd):
    """True when head dtype `hd` runs the row-chain kernels ('bf16' | 'fp16')."""
    return hd in ('bf16', 'fp16')


class HeadPack:
    """bf16 operand copies (fragment-major images of W and of W^T) of a set of nn.Linear weights, refreshed by ONE launch when any source
    changed: torch's version counters catch in-place writes (load_state_dict, torch optimizers), `invalidate()` is what the
    fused optimizer calls after it updated the parameters through raw pointers (FusedAdam.packed_owners).
    One instance may serve several modules (models.transformer.TransformerModel shares one across the head): every caller's weights
    stay registered under its `tag`, and whichever caller finds the pack stale refreshes ALL of them in the same launch."""

    def __init__(self):
        self.key = None
        self.buf = None
        self.views = {}
        self.stale = True
        self.registry = {}            # tag -> [(name, weight)] as last seen
        self.f16 = False              # MI355X.HEAD_DTYPE fp16: the FORWARD images (W) are IEEE fp16, the transposed ones stay bf16

    def set_f16(self, on):
        if bool(on) != self.f16:
            self.f16, self.stale = bool(on), True

    def invalidate(self):
        self.stale = True

    @staticmethod
    def _key(named):
        return tuple((n, w.data_ptr(), w._version, tuple(w.shape), w.stride(0)) for n, w in named)

    def get(self, named, tag=''):
        """named: [(name, weight [N, K] fp32 with unit inner stride)] -> {name: (w16 ptr, w16t ptr)}"""
        named = [(tag + n, w) for n, w in named]
        old = self.registry.get(tag)
        self.registry[tag] = named
        allw = [nw for t in sorted(self.registry) for nw in self.registry[t]]
        key = self._key(allw)
        if self.stale or key != self.key or old is None:
            self._pack(allw)
            self.key, self.stale = key, False
        return {n[len(tag):]: self.views[n] for n, _w in named}

    def _pack(self, named):
        lib = _lib.load()
        dev = named[0][1].device
        sizes = [(lib.mvf_head_pack_elems(w.shape[0], w.shape[1], 0), lib.mvf_head_pack_elems(w.shape[0], w.shape[1], 1))
                 for _n, w in named]
        total = sum((a + 63) // 64 * 64 + (b + 63) // 64 * 64 for a, b in sizes)
        if self.buf is None or self.buf.numel() < total or self.buf.device != dev:
            self.buf = torch.empty(total, device=dev, dtype=torch.bfloat16)
        ents = (_lib.MvfPackEntry * len(named))()
        views, off, base = {}, 0, self.buf.data_ptr()
        for i, ((n, w), (a, b)) in enumerate(zip(named, sizes)):
            if not w.is_cuda:
                raise _lib.MvfError('HIP op received a %s tensor (no CPU fallback)' % w.device)
            assert w.dim() == 2 and w.dtype == torch.float32 and w.stride(1) == 1
            p16, off = base + 2 * off, off + (a + 63) // 64 * 64
            p16t, off = base + 2 * off, off + (b + 63) // 64 * 64
            e = ents[i]
            e.w, e.ld, e.N, e.K, e.w16, e.w16t, e.f16 = w.data_ptr(), w.stride(0), w.shape[0], w.shape[1], p16, p16t, int(self.f16)
            views[n] = (p16, p16t)
        for i0 in range(0, len(named), 32):
            n = min(32, len(named) - i0)
            call('mvf_head_pack_weights', ctypes.byref(ents, i0 * ctypes.sizeof(_lib.MvfPackEntry)), n, stream())
        self.views = views


def _drop_c(d):
    """(p, seed, offset) | None -> MvfDrop"""
    m = _lib.MvfDrop()
    if d is not None:
        m.p, m.seed, m.offset = float(d[0]), int(d[1]), int(d[2])
    return m


def encoder_chain_supported(D, DFF, H):
    return HEAD_CHAIN and D % 256 == 0 and DFF % 256 == 0 and D <= 512 and D % H == 0 and (D // H) in (16, 32, 64) and \
        32 * ((D + 4) * 8 + (D + 8) * 2 + (max(DFF, 3 * D) + 8) * 2) <= 160 * 1024


class _EncoderChain(torch.autograd.Function):
    """The temporal Encoder (models/utils.py:228-242: N pre-LN EncoderLayers, no final norm) as ONE autograd node on the
    row-chain kernels: per layer one attention launch + one chain launch forward; one chain launch + the two attention-backward
    launches backward; all weight / bias gradients of the encoder in one launch at the end.
    params per layer (12): ln0.w, ln0.b, Wqkv [3D, D], bqkv, Wo, bo, ln1.w, ln1.b, W1, b1, W2, b2."""

    NP = 12

    @staticmethod
    def forward(ctx, x, mask, B, S, H, eps, drops, pack, slots, owners, *params):
        L = len(params) // _EncoderChain.NP
        x = x.contiguous()
        M, D = x.shape
        DFF = params[8].shape[0]
        dev = x.device
        Mp = (M + 127) // 128 * 128
        need = any(ctx.needs_input_grad)
        named = []
        for l in range(L):
            P = params[l * 12:(l + 1) * 12]
            named += [('qkv%d' % l, P[2]), ('o%d' % l, P[4]), ('f1%d' % l, P[8]), ('f2%d' % l, P[10])]
        W = pack.get(named, 'enc.')
        mlen, mk = S, None
        if mask is not None:
            mk = mask.reshape(B, -1)
            mlen = mk.shape[1]
            assert S % mlen == 0, (S, mlen)
            if mk.dtype != torch.float32 or not mk.is_contiguous():
                mk = mk.contiguous().float()

        def f32(*shape):
            return torch.empty(*shape, device=dev, dtype=torch.float32)

        def b16(rows, cols):
            """bf16 [rows, cols] (the row save of `a`), or -- cols == Mp -- the fragment-major image of a transposed operand
            (rows padded to 64: include/mvf_hip.h, "FM")"""
            return torch.empty((rows + 63) // 64 * 64, cols, device=dev, dtype=torch.bfloat16) if need else None

        saved = []
        xs = [x]
        a = _lib.MvfEncFwd()
        a.M, a.D, a.DFF, a.Mp, a.ln_eps, a.f16 = M, D, DFF, Mp, eps, int(pack.f16)
        P0 = params[0:12]
        qkv, mean0, rstd0, h0T = f32(M, 3 * D), f32(M), f32(M), b16(D, Mp)
        a.x_in, a.wqkv, a.bqkv, a.ln0_g, a.ln0_b = ptr(x), W['qkv0'][0], ptr(P0[3]), ptr(P0[0]), ptr(P0[1])
        a.qkv, a.mean0, a.rstd0, a.h0T = ptr(qkv), ptr(mean0), ptr(rstd0), ptr(h0T)
        call('mvf_enc_layer_fwd', ctypes.byref(a), stream())
        for l in range(L):
            P = params[l * 12:(l + 1) * 12]
            o, lse = f32(M, D), f32(B, H, S)
            call('mvf_tattn_fwd', ptr(qkv), ptr(mk), mlen, ptr(o), ptr(lse), B, S, H, D, stream())
            x1, mean1, rstd1, x2 = f32(M, D), f32(M), f32(M), f32(M, D)
            act, oT, h1T, aT = b16(M, DFF), b16(D, Mp), b16(D, Mp), b16(DFF, Mp)
            a = _lib.MvfEncFwd()
            a.M, a.D, a.DFF, a.Mp, a.ln_eps, a.f16 = M, D, DFF, Mp, eps, int(pack.f16)
            a.o, a.x_in = ptr(o), ptr(xs[-1])
            a.wo, a.w1, a.w2 = W['o%d' % l][0], W['f1%d' % l][0], W['f2%d' % l][0]
            a.bo, a.b1, a.b2, a.ln1_g, a.ln1_b = ptr(P[5]), ptr(P[9]), ptr(P[11]), ptr(P[6]), ptr(P[7])
            a.drop_attn, a.drop_ffn = _drop_c(drops[2 * l]), _drop_c(drops[2 * l + 1])
            a.x1, a.mean1, a.rstd1, a.a, a.x2 = ptr(x1), ptr(mean1), ptr(rstd1), ptr(act), ptr(x2)
            a.oT, a.h1T, a.aT = ptr(oT), ptr(h1T), ptr(aT)
            rec = dict(qkv=qkv, o=o, lse=lse, x1=x1, mean1=mean1, rstd1=rstd1, a=act, oT=oT, h1T=h1T, aT=aT, mean0=mean0, rstd0=rstd0,
                       h0T=h0T)
            if l + 1 < L:
                Pn = params[(l + 1) * 12:(l + 2) * 12]
                qkv, mean0, rstd0, h0T = f32(M, 3 * D), f32(M), f32(M), b16(D, Mp)
                a.wqkv, a.bqkv, a.ln0_g, a.ln0_b = W['qkv%d' % (l + 1)][0], ptr(Pn[3]), ptr(Pn[0]), ptr(Pn[1])
                a.qkv, a.mean0, a.rstd0, a.h0T = ptr(qkv), ptr(mean0), ptr(rstd0), ptr(h0T)
            call('mvf_enc_layer_fwd', ctypes.byref(a), stream())
            saved.append(rec)
            xs.append(x2)
        if need:
            ctx.saved = saved
            ctx.xs = xs
            ctx.params = params
            ctx.cfg = (B, S, H, M, D, DFF, Mp, L, mk, mlen, drops, W, slots, owners)
            ctx.keep = pack.buf         # the bf16 copies the backward reads (a later refresh allocates a new buffer only if it grows)
        return xs[-1]

    @staticmethod
    def backward(ctx, dy):
        B, S, H, M, D, DFF, Mp, L, mk, mlen, drops, W, slots, owners = ctx.cfg
        params, saved, xs = ctx.params, ctx.saved, ctx.xs
        dev = dy.device
        dy = dy.contiguous().view(M, D)

        def f32(*shape):
            return torch.empty(*shape, device=dev, dtype=torch.float32)

        def b16(rows, cols):
            return torch.empty((rows + 63) // 64 * 64, cols, device=dev, dtype=torch.bfloat16)

        use_slots = slots is not None
        grads = [None] * len(params)

        def gbuf(i, zero):
            """gradient buffer of params[i]: its flat-gradient slot, or a fresh tensor"""
            if use_slots:
                return slots[i]
            g = (torch.zeros_like if zero else torch.empty_like)(params[i])
            grads[i] = g
            return g

        probs = []
        keep = []
        dres, dqkv = dy, None
        for l in range(L - 1, -2, -1):
            a = _lib.MvfEncBwd()
            a.M, a.D, a.DFF, a.Mp = M, D, DFF, Mp
            a.dres = ptr(dres)
            if dqkv is not None:        # segment B' of layer l + 1
                lb = l + 1
                R = saved[lb]
                dqkvT = b16(3 * D, Mp)
                a.dqkv, a.wqkvT, a.x_in = ptr(dqkv), W['qkv%d' % lb][1], ptr(xs[lb])
                a.mean0, a.rstd0, a.ln0_g = ptr(R['mean0']), ptr(R['rstd0']), ptr(params[lb * 12 + 0])
                a.dln0_g, a.dln0_b = gbuf(lb * 12 + 0, True).data_ptr(), gbuf(lb * 12 + 1, True).data_ptr()
                a.dqkvT = ptr(dqkvT)
                probs.append((dqkvT, R['h0T'], lb * 12 + 2, lb * 12 + 3, 3 * D, D))
                keep.append(dqkvT)
            dx_out = None
            if l < 0:
                dx_out = f32(M, D)
                a.dx_out = ptr(dx_out)
            else:                        # segment A' of layer l
                R = saved[l]
                g2T, duT, goT, dx1, d_o = b16(D, Mp), b16(DFF, Mp), b16(D, Mp), f32(M, D), f32(M, D)
                a.drop_attn, a.drop_ffn = _drop_c(drops[2 * l]), _drop_c(drops[2 * l + 1])
                a.w2T, a.w1T, a.woT = W['f2%d' % l][1], W['f1%d' % l][1], W['o%d' % l][1]
                a.a, a.x1, a.mean1, a.rstd1, a.ln1_g = ptr(R['a']), ptr(R['x1']), ptr(R['mean1']), ptr(R['rstd1']), ptr(params[l * 12 + 6])
                a.dln1_g, a.dln1_b = gbuf(l * 12 + 6, True).data_ptr(), gbuf(l * 12 + 7, True).data_ptr()
                a.g2T, a.duT, a.goT, a.dx1_out, a.d_o = ptr(g2T), ptr(duT), ptr(goT), ptr(dx1), ptr(d_o)
                probs += [(g2T, R['aT'], l * 12 + 10, l * 12 + 11, D, DFF), (duT, R['h1T'], l * 12 + 8, l * 12 + 9, DFF, D),
                          (goT, R['oT'], l * 12 + 4, l * 12 + 5, D, D)]
                keep += [g2T, duT, goT]
            call('mvf_enc_layer_bwd', ctypes.byref(a), stream())
            if l < 0:
                break
            dqkv = f32(M, 3 * D)
            call('mvf_tattn_bwd', ptr(R['qkv']), ptr(mk), mlen, ptr(R['o']), ptr(R['lse']), ptr(d_o), ptr(dqkv), B, S, H, D, stream())
            dres = dx1
        # every weight / bias gradient of the encoder: one launch per 16 Linears
        for i0 in range(0, len(probs), 16):
            chunk = probs[i0:i0 + 16]
            arr = (_lib.MvfDwProblem * len(chunk))()
            for e, (gT, xT, iw, ib, N, K) in zip(arr, chunk):
                gw, gb = gbuf(iw, False), gbuf(ib, False)
                e.gT, e.xT, e.dw, e.lddw, e.db, e.N, e.K = ptr(gT), ptr(xT), gw.data_ptr(), gw.stride(0) if gw.dim() == 2 else K, gb.data_ptr(), N, K
            call('mvf_head_dw', arr, len(chunk), Mp, 1 if use_slots else 0, stream())
        if use_slots:
            grad_ready(*owners)
        ctx.saved = ctx.xs = None
        return (dx_out if ctx.needs_input_grad[0] else None, None, None, None, None, None, None, None, None, None) + tuple(grads)


class RowLinStage:
    """One Linear of a row chain (csrc/head_rowlin.hip mvf_rowlin_fwd) and what surrounds it.  Parameter fields hold INDICES into
    the chain's flat parameter list (None: absent)."""

    def __init__(self, w, b, gather=None, bn_in=None, onehot=None, drop_in=None, table=None, drop_out=None, l2norm=None, bn_out=None,
                 sync=None):
        self.w, self.b = w, b
        self.gather = gather          # (ntok, T, mode 0 | 1 | 2): entity reduction of the input rows
        self.bn_in = bn_in            # the BatchNorm (+ReLU) in front: (index of gamma, index of beta, eps, relu); its statistics come
        #                               from the previous stage's bn_out (training) or from `running` (eval)
        self.onehot = onehot          # (ntok, div)
        self.drop_in, self.drop_out = drop_in, drop_out      # (p, seed, offset) | None
        self.table = table            # (tensor [mod, N], mod)
        self.l2norm = l2norm          # eps | None
        self.bn_out = bn_out          # the BatchNorm that follows: (running_mean, running_var, momentum) buffers (or Nones)
        self.sync = sync              # (process group | None,): bn_out is a SyncBatchNorm with live collectives -- its statistics (forward)
        #                               and its backward sums are exchanged between this stage's launch and the next one's


def rowlin_supported(n_in, n_out):
    return HEAD_CHAIN and n_in % 4 == 0 and n_in <= 512 and n_out % 128 == 0 and n_out <= 512


class _RowLinChain(torch.autograd.Function):
    """A run of Linear stages with BatchNorms between them as one autograd node: one launch per stage each way (BatchNorm statistics,
    application, ReLU, dropout, one-hot, entity reduction, normalisation all inside), one weight-gradient launch at the end."""

    @staticmethod
    def forward(ctx, x, stages, training, pack, slots, owners, eval_stats, tag, *params):
        dev = x.device
        need = any(ctx.needs_input_grad)
        x = x.contiguous()
        W = pack.get([('s%d' % i, params[st.w]) for i, st in enumerate(stages)], tag)
        cur = x.view(-1, x.shape[-1])
        rec = []
        stats = None                   # (mean, var) of the BatchNorm in front of the next stage
        in_count = 0.0                 # rows behind those statistics (all ranks' rows for a SyncBatchNorm)
        for i, st in enumerate(stages):
            w = params[st.w]
            N, Kin = w.shape
            rows_in, Cin = cur.shape
            M = rows_in // st.gather[0] if st.gather else rows_in
            Mp = (M + 127) // 128 * 128
            G = (M + 31) // 32
            a = _lib.MvfRowLinFwd()
            a.M, a.Cin, a.N, a.Mp, a.X, a.ldx = M, Cin, N, Mp, ptr(cur), cur.stride(0)
            g_arg = None
            if st.gather:
                a.g_ntok, a.g_T, a.g_mode = st.gather
                if st.gather[2] == 2:
                    g_arg = torch.empty(M, Cin, device=dev, dtype=torch.int32)
                    a.g_arg = ptr(g_arg)
            bn_stats = None
            if st.bn_in is not None:
                ig, ib, eps, relu = st.bn_in
                bn_stats = stats if training else eval_stats[i]
                a.bn_mean, a.bn_var, a.bn_g, a.bn_b, a.bn_eps, a.bn_relu = ptr(bn_stats[0]), ptr(bn_stats[1]), ptr(params[ig]), ptr(params[ib]), eps, int(relu)
            if st.onehot:
                a.oh_ntok, a.oh_div = st.onehot
            assert Cin + (st.onehot[0] if st.onehot else 0) == Kin, (Cin, Kin)
            a.drop_in, a.drop_out = _drop_c(st.drop_in), _drop_c(st.drop_out)
            a.w16, a.bias, a.f16 = W['s%d' % i][0], ptr(params[st.b]) if st.b is not None else None, int(pack.f16)
            if st.table is not None:
                a.table, a.tab_mod = ptr(st.table[0]), st.table[1]
            Y = torch.empty(M, N, device=dev, dtype=torch.float32)
            nrm = None
            if st.l2norm is not None:
                nrm = torch.empty(M, device=dev, dtype=torch.float32)
                a.l2norm, a.l2_eps, a.nrm = 1, st.l2norm, ptr(nrm)
            a.Y = ptr(Y)
            xT = torch.empty((Kin + 63) // 64 * 64, Mp, device=dev, dtype=torch.bfloat16) if need else None
            a.xT = ptr(xT)
            stats, out_count = None, float(M)
            synced = st.sync is not None and st.bn_out is not None and training
            if st.bn_out is not None and training:
                part = torch.empty(2 * G * N, device=dev, dtype=torch.float32)
                sync_local = syncbn_local_buffer(N, float(M), dev) if synced else None   # SyncBatchNorm: this rank's block of the exchange
                if synced:
                    mean, var = sync_local[:N], sync_local[N:2 * N]
                else:
                    mean, var = torch.empty(N, device=dev, dtype=torch.float32), torch.empty(N, device=dev, dtype=torch.float32)
                a.st_part, a.st_mean, a.st_var = ptr(part), ptr(mean), ptr(var)
                rm, rv, mom = st.bn_out
                if rm is not None and not synced:      # (SyncBatchNorm: the running buffers take the merged statistics, below)
                    a.st_rmean, a.st_rvar, a.st_momentum = ptr(rm), ptr(rv), float(mom)
                stats = (mean, var)
            call('mvf_rowlin_fwd', ctypes.byref(a), stream())
            if synced:
                # SyncBatchNorm: (mean, biased var, count) of every rank merged (Chan) between this launch and the next -- collective
                # C2 of SURVEY.md, 2N + 1 floats per rank
                rm, rv, mom = st.bn_out
                mean, var, out_count = sync_bn_stats(stats[0], stats[1], float(M), st.sync[0], equal_counts=True,
                                                     running=(rm, rv, mom) if rm is not None else None, local=sync_local)
                stats = (mean, var)
            rec.append(dict(X=cur, Y=Y, bn=bn_stats, bn_count=in_count, xT=xT, nrm=nrm, g_arg=g_arg, M=M, Mp=Mp, Cin=Cin, Kin=Kin, N=N))
            in_count = out_count
            cur = Y
        if need:
            ctx.rec, ctx.stages, ctx.params, ctx.W, ctx.slots, ctx.owners, ctx.training = rec, stages, params, W, slots, owners, training
            ctx.keep = pack.buf
            ctx.xshape = x.shape
        return cur

    @staticmethod
    def backward(ctx, dy):
        rec, stages, params, W, slots, owners, training = ctx.rec, ctx.stages, ctx.params, ctx.W, ctx.slots, ctx.owners, ctx.training
        dev = dy.device
        use_slots = slots is not None
        grads = [None] * len(params)

        def gbuf(i, zero):
            if use_slots:
                return slots[i]
            if grads[i] is None:
                grads[i] = (torch.zeros_like if zero else torch.empty_like)(params[i])
            return grads[i]

        dcur = dy.contiguous().view(rec[-1]['M'], rec[-1]['N'])
        nb = None
        probs, keep = [], []
        for i in range(len(stages) - 1, -1, -1):
            st, R = stages[i], rec[i]
            M, Cin, Kin, N, Mp = R['M'], R['Cin'], R['Kin'], R['N'], R['Mp']
            G = (M + 31) // 32
            a = _lib.MvfRowLinBwd()
            a.M, a.Cin, a.N, a.Mp, a.dY = M, Cin, N, Mp, ptr(dcur)
            if nb is not None:          # the BatchNorm that consumes this stage's Y: its backward, applied while dZ is loaded
                a.nb_Y, a.nb_mean, a.nb_var, a.nb_g, a.nb_s1, a.nb_s2, a.nb_eps, a.nb_count = nb
            a.drop_out, a.drop_in = _drop_c(st.drop_out), _drop_c(st.drop_in)
            if st.l2norm is not None:
                a.l2norm, a.l2_y, a.l2_nrm, a.l2_eps = 1, ptr(R['Y']), ptr(R['nrm']), st.l2norm
            gT = torch.empty((N + 63) // 64 * 64, Mp, device=dev, dtype=torch.bfloat16)
            a.w16t, a.gT = W['s%d' % i][1], ptr(gT)
            a.oh_ntok = st.onehot[0] if st.onehot else 0
            nb, sync_sums = None, None
            if st.bn_in is not None:
                ig, ib, eps, relu = st.bn_in
                mean, var = R['bn']
                a.X, a.ldx = ptr(R['X']), R['X'].stride(0)
                a.bn_mean, a.bn_var, a.bn_g, a.bn_b, a.bn_eps, a.bn_relu = ptr(mean), ptr(var), ptr(params[ig]), ptr(params[ib]), eps, int(relu)
                part = torch.empty(2 * G * Cin, device=dev, dtype=torch.float32)
                s12 = torch.empty(2, Cin, device=dev, dtype=torch.float32)
                s1, s2 = s12[0], s12[1]
                a.st_part, a.s1, a.s2 = ptr(part), ptr(s1), ptr(s2)
                a.dgamma, a.dbeta = gbuf(ig, True).data_ptr(), gbuf(ib, True).data_ptr()
                nb = (ptr(R['X']), ptr(mean), ptr(var), ptr(params[ig]), ptr(s1), ptr(s2), eps, R['bn_count'] if training else 0.0)
                keep += [part, s12]
                sync_sums = s12 if (training and i > 0 and stages[i - 1].sync is not None) else None
            rows_in = R['X'].shape[0]
            dX = torch.empty(rows_in, Cin, device=dev, dtype=torch.float32)
            if st.gather:
                a.g_ntok, a.g_T, a.g_mode = st.gather
                a.g_arg = ptr(R['g_arg'])
            a.dX, a.lddx = ptr(dX), Cin
            call('mvf_rowlin_bwd', ctypes.byref(a), stream())
            if sync_sums is not None:
                # SyncBatchNorm backward: sum dZ and sum dZ xhat over ALL ranks' rows (collective C3 of SURVEY.md) before the previous
                # stage's launch applies them; dgamma / dbeta stay local (the gradient all-reduce averages them like every parameter)
                dist.all_reduce(sync_sums, group=stages[i - 1].sync[0])
            probs.append((gT, R['xT'], st.w, st.b, N, Kin))
            keep.append(gT)
            dcur = dX
        for i0 in range(0, len(probs), 16):
            chunk = probs[i0:i0 + 16]
            arr = (_lib.MvfDwProblem * len(chunk))()
            mp = None
            for e, (gT, xT, iw, ib, N, K) in zip(arr, chunk):
                gw = gbuf(iw, False)
                e.gT, e.xT, e.dw, e.lddw, e.N, e.K = ptr(gT), ptr(xT), gw.data_ptr(), gw.stride(0) if gw.dim() == 2 else K, N, K
                e.db = gbuf(ib, False).data_ptr() if ib is not None else None
                mp = gT.shape[1] if mp is None else mp
                assert gT.shape[1] == mp
            call('mvf_head_dw', arr, len(chunk), mp, 1 if use_slots else 0, stream())
        if use_slots:
            grad_ready(*owners)
        ctx.rec = None
        dx = dcur.view(ctx.xshape) if ctx.needs_input_grad[0] else None
        return (dx, None, None, None, None, None, None, None) + tuple(grads)


def rowlin_chain(x, stages, params, training, pack, eval_stats=None, tag=''):
    """x [..., C] through the stages (RowLinStage, indices into `params`); parameter gradients go straight into their flat-gradient
    slots when every parameter has one.  eval_stats[i] = (running_mean, running_var) of stage i's bn_in (eval mode)."""
    slots = [grad_slot(p) for p in params]
    use_slots = x.requires_grad and all(s_ is not None for s_ in slots)
    return _RowLinChain.apply(x, tuple(stages), bool(training), pack, tuple(slots) if use_slots else None, tuple(params) if use_slots else (),
                              eval_stats, tag, *params)


_SPARE_SET = [None]
_SPARE_MODE = [True]


def gemm_spare_mode(on):
    """on: the backbone's forwards run BESIDE head work (training with the one-batch lookahead: models/transformer.py) and the persistent GEMM
    leaves spare CUs for it; off (evaluation: forward and head one after the other): one workgroup per CU, the spare would only cost."""
    on = bool(on)
    if _SPARE_MODE[0] != on:
        _SPARE_MODE[0] = on
        if os.environ.get('MVF_GEMM_SPARE') is None:
            want = 32 if on else 0            # (on: the head's row count refines it, _spare_cus_for_rows)
            if _SPARE_SET[0] != want:
                call('mvf_gemm_tc_set_spare', want)
                _SPARE_SET[0] = want


def _spare_cus_for_rows(rows):
    """The backbone's persistent GEMM leaves the CUs free that the head's row-chain launches (one workgroup per 32 rows) will hold beside
    it (include/mvf_hip.h mvf_gemm_tc_set_spare; measured at 24 workgroups: 32 spare CUs -0.15 .. -0.28 ms per step, fewer no gain,
    more no further gain): 32 up to 1 024 rows, the workgroup count rounded up to a multiple of 8 beyond, 64 at most.  MVF_GEMM_SPARE pins it."""
    if os.environ.get('MVF_GEMM_SPARE') is not None:
        return
    want = max(32, min(64, ((rows + 31) // 32 + 7) // 8 * 8)) if _SPARE_MODE[0] else 0
    if _SPARE_SET[0] != want:
        call('mvf_gemm_tc_set_spare', want)
        _SPARE_SET[0] = want


def encoder_chain(x, mask, layers, H, eps, drops, pack):
    """x [B, S, D] -> [B, S, D].  layers: per EncoderLayer a dict with the 12 parameter tensors in _EncoderChain's order
    ('params'), and for parameters that live in the flat gradient buffer 'slots' (12 gradient views, the Q|K|V one over the
    three projections) + 'owners' (the nn.Parameters to report ready); drops: (p, seed, offset) | None per sub-layer
    (attention, feed-forward) in layer order."""
    B, S, D = x.shape
    _spare_cus_for_rows(B * S)
    params, slots, owners = [], [], []
    use_slots = x.requires_grad and all(ly.get('slots') is not None for ly in layers)
    for ly in layers:
        params += list(ly['params'])
        if use_slots:
            slots += list(ly['slots'])
            owners += list(ly['owners'])
    if use_slots:
        for o in owners:
            grad_slot(o)        # marks the flat gradient buffer as written
    y = _EncoderChain.apply(x.reshape(B * S, D), mask, B, S, H, float(eps), tuple(drops), pack, tuple(slots) if use_slots else None,
                            tuple(owners), *params)
    return y.view(B, S, D)


# ------------------------------------------------------------------------------------------------
# small row ops
# ------------------------------------------------------------------------------------------------
class _ConcatOneHot(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ntok, div):
        x = x.contiguous()
        R, cin = x.shape
        out = torch.empty(R, cin + ntok, device=x.device, dtype=torch.float32)
        call('mvf_concat_onehot', ptr(x), ptr(out), R, cin, ntok, div, stream())
        ctx.cin = cin
        return out

    @staticmethod
    def backward(ctx, dout):
        return dout[:, :ctx.cin], None, None


def concat_onehot(x, ntok, div):
    """[R, C] -> [R, C + ntok] with one-hot of entity id (row // div) % ntok appended."""
    return _ConcatOneHot.apply(x, ntok, div)


_REDUCE_MODES = {'one': 0, 'avg': 1, 'max': 2}


class _FinalReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        x = x.contiguous()
        B, ntok, T, D = x.shape
        y = torch.empty(B, T, D, device=x.device, dtype=torch.float32)
        arg = torch.empty(B, T, D, device=x.device, dtype=torch.int32) if mode == 2 else None
        call('mvf_final_reduce_fwd', ptr(x), ptr(y), ptr(arg), B, ntok, T, D, mode, stream())
        ctx.save_for_backward(arg)
        ctx.cfg = (B, ntok, T, D, mode)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        B, ntok, T, D, mode = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty(B, ntok, T, D, device=dy.device, dtype=torch.float32)
        call('mvf_final_reduce_bwd', ptr(dy), ptr(arg), ptr(dx), B, ntok, T, D, mode, stream())
        return dx, None


def final_reduce(x, mode):
    """x [B, ntok, T, D] -> [B, T, D]; mode in {'one','avg','max'} (SMART_FINAL)."""
    return _FinalReduce.apply(x, _REDUCE_MODES[mode])


class _L2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        x = x.contiguous()
        R, D = x.shape
        y = torch.empty_like(x)
        nrm = torch.empty(R, device=x.device, dtype=torch.float32)
        call('mvf_l2norm_fwd', ptr(x), ptr(y), ptr(nrm), R, D, eps, stream())
        ctx.save_for_backward(y, nrm)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        y, nrm = ctx.saved_tensors
        dy = dy.contiguous()
        R, D = y.shape
        dx = torch.empty_like(y)
        call('mvf_l2norm_bwd', ptr(dy), ptr(y), ptr(nrm), ptr(dx), R, D, ctx.eps, stream())
        return dx, None


def l2_normalize(x, eps=1e-12):
    shp = x.shape
    return _L2Norm.apply(x.reshape(-1, shp[-1]), eps).view(shp)


class _DropoutAdd(torch.autograd.Function):
    """y = resid + dropout_p(x); the mask is a pure function of (seed, offset, index) and is regenerated in backward."""

    @staticmethod
    def forward(ctx, x, resid, p, seed, offset):
        x = x.contiguous()
        if resid is not None:
            resid = resid.contiguous()
        y = torch.empty_like(x)
        call('mvf_dropout_add', ptr(x), ptr(resid), ptr(y), x.numel(), p, seed, offset, stream())
        ctx.cfg = (p, seed, offset, resid is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed, offset, has_resid = ctx.cfg
        dy = dy.contiguous()
        dx = dy
        if p > 0.0:
            dx = torch.empty_like(dy)
            call('mvf_dropout_add', ptr(dy), None, ptr(dx), dy.numel(), p, seed, offset, stream())
        return dx, (dy if has_resid else None), None, None, None


class DropoutState:
    """Seed + running offset of the counter-based dropout masks (one per model; advanced per call)."""

    def __init__(self, seed=0):
        self.seed = int(seed) & 0x7FFFFFFFFFFFFFFF
        self.offset = 0

    def next(self, n):
        o = self.offset
        self.offset += n
        return self.seed, o


def drop_args(p, training, state, n):
    """(p, seed, offset) for a dropout fused into a GEMM (ops.linear `drop=`), or None when it is the identity."""
    p = float(p) if training else 0.0
    if p == 0.0:
        return None
    seed, off = state.next(n)
    return (p, seed, off)


def dropout_add(x, resid, p, training, state):
    """resid + dropout(x) (resid may be None).  With p == 0 or eval and no residual this is the identity."""
    p = float(p) if training else 0.0
    if resid is None and p == 0.0:
        return x
    seed, off = state.next(x.numel()) if p > 0.0 else (0, 0)
    return _DropoutAdd.apply(x, resid, p, seed, off)


# ------------------------------------------------------------------------------------------------
# LSTP pooling core over the backbone taps
# ------------------------------------------------------------------------------------------------
_CONST = {}


def _ones(shape, device):
    """A cached all-ones fp32 tensor (read-only by convention: returned to callers as a non-differentiable output)."""
    key = (tuple(shape), str(device))
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.ones(tuple(shape), device=device, dtype=torch.float32)
    return t


# MVF_LSTP_ONE_PASS=0: the three-launch pooling (scores / softmax / weighted sum) everywhere (A/B measurements, tests)
LSTP_ONE_PASS = os.environ.get('MVF_LSTP_ONE_PASS', '1') != '0'


def _tap_table(taps):
    arr = (ctypes.c_void_p * len(taps))(*[ptr(t) for t in taps])
    return arr


class _LSTPPool(torch.autograd.Function):
    """pooled[b, j, t, :] = sum_n A[f,j,n] * x[f,n,:],  A = softmax_n(x[f,n,:] . vec[j,:] / sqrt(d))   (f = b*T + t)
    (disjoint: A masked to the arg-max query per token, models/utils.py:26-33).
    `vec` is [nq, C] (static queries) or [Bc, nq, T, C] (per-frame / dynamic queries).
    Returns (pooled [Bc, nq, T, C], rowsum [Bc, nq, T] = sum_n A -- identically 1 unless disjoint)."""

    @staticmethod
    def forward(ctx, vec, taps, F, N, T, nq, inv_sqrt_d, disjoint, holder, *grad_taps):
        # grad_taps: the same tensors again as differentiable inputs when the tapped blocks are trainable (fp32 taps)
        ctx.want_dx = len(grad_taps) > 0 and any(t.requires_grad for t in grad_taps)
        dt = BF16 if taps[0].dtype == torch.bfloat16 else F32
        D = taps[0].shape[1]
        C = D * len(taps)
        dev = taps[0].device
        vec = vec.contiguous()
        per_frame = vec.dim() == 4
        tab = _tap_table(taps)
        P = torch.empty(F, nq, N, device=dev, dtype=torch.float32)
        pooled = torch.empty(F // T, nq, T, C, device=dev, dtype=torch.float32)
        Pm = rowsum = None
        # one pass over the taps (online softmax) where the kernel's register budget allows it: nq <= 3, <= 3 taps, D <= 1024
        fused = LSTP_ONE_PASS and not disjoint and not ctx.want_dx and \
            _lib.try_call('mvf_lstp_fused_fwd', tab, len(taps), dt, D, F, N, T, nq, ptr(vec), int(per_frame), inv_sqrt_d, ptr(P),
                          ptr(pooled), stream())
        if not fused:
            scores = torch.empty(F * N, nq, device=dev, dtype=torch.float32)
            call('mvf_lstp_scores', tab, len(taps), dt, D, F, N, T, nq, ptr(vec), int(per_frame), ptr(scores), stream())
            Pm = torch.empty_like(P) if disjoint else None
            rowsum = torch.empty(F, nq, device=dev, dtype=torch.float32) if disjoint else None
            call('mvf_lstp_softmax_fwd', ptr(scores), ptr(P), ptr(Pm), ptr(rowsum), F, N, nq, inv_sqrt_d, int(disjoint),
                 stream())
            call('mvf_lstp_wsum', tab, len(taps), dt, D, F, N, T, nq, ptr(Pm if disjoint else P), ptr(pooled), stream())
        ctx.taps = taps
        ctx.fused = fused
        ctx.save_for_backward(P, Pm, vec if ctx.want_dx else None, pooled if fused else None)
        ctx.cfg = (F, N, T, nq, inv_sqrt_d, per_frame, dt, D, C)
        if holder is not None:
            holder['attn'] = Pm if disjoint else P      # [F, nq, N], like LSTPCrossAtt.attn_matrix
        ctx.set_materialize_grads(False)       # no zero-filled gradient for the (constant) row sums
        if disjoint:
            rs = rowsum.view(F // T, T, nq).transpose(1, 2).contiguous()     # (b, j, t) like pooled
        else:
            rs = _ones((F // T, nq, T), dev).view(F // T, nq, T)
            ctx.mark_non_differentiable(rs)
        return pooled, rs

    @staticmethod
    def backward(ctx, dpooled, drs):
        P, Pm, vec, pooled = ctx.saved_tensors
        if dpooled is None:
            dpooled = torch.zeros(ctx.cfg[0] // ctx.cfg[2], ctx.cfg[3], ctx.cfg[2], ctx.cfg[8], device=P.device, dtype=torch.float32)
        F, N, T, nq, inv_sqrt_d, per_frame, dt, D, C = ctx.cfg
        taps = ctx.taps
        tab = _tap_table(taps)
        dev = dpooled.device
        dpooled = dpooled.contiguous()
        if ctx.fused:         # one pass: G = d loss / d vec per frame from dpooled, P and the forward's pooled
            G = torch.empty(F // T, nq, T, C, device=dev, dtype=torch.float32)
            call('mvf_lstp_fused_bwd', tab, len(taps), dt, D, F, N, T, nq, ptr(dpooled), ptr(P), ptr(pooled), inv_sqrt_d, ptr(G),
                 stream())
            if per_frame:
                dvec = G
            else:
                dvec = torch.empty(nq, C, device=dev, dtype=torch.float32)
                call('mvf_lstp_reduce_frames', ptr(G), ptr(dvec), F // T, nq, T, C, stream())
            return (dvec, None, None, None, None, None, None, None, None)
        dP = torch.empty(F * N, nq, device=dev, dtype=torch.float32)
        call('mvf_lstp_scores', tab, len(taps), dt, D, F, N, T, nq, ptr(dpooled), 1, ptr(dP), stream())
        drow = None
        if Pm is not None and drs is not None:
            drow = drs.transpose(1, 2).contiguous().view(F, nq)             # back to (f, j)
        dS = torch.empty(F, nq, N, device=dev, dtype=torch.float32)
        call('mvf_lstp_softmax_bwd', ptr(P), ptr(Pm), ptr(dP), ptr(drow), ptr(dS), F, N, nq, inv_sqrt_d, stream())
        G = torch.empty(F // T, nq, T, C, device=dev, dtype=torch.float32)
        call('mvf_lstp_wsum', tab, len(taps), dt, D, F, N, T, nq, ptr(dS), ptr(G), stream())
        if per_frame:
            dvec = G
        else:
            dvec = torch.empty(nq, C, device=dev, dtype=torch.float32)
            call('mvf_lstp_reduce_frames', ptr(G), ptr(dvec), F // T, nq, T, C, stream())
        dtaps = ()
        if ctx.want_dx:
            if dt != F32:
                raise _lib.MvfError('token gradients need fp32 taps (trainable backbone blocks run in fp32)')
            dtaps = tuple(torch.empty_like(t) for t in taps)
            arr = (ctypes.c_void_p * len(taps))(*[t.data_ptr() for t in dtaps])
            call('mvf_lstp_dx', arr, len(taps), D, F, N, T, nq, ptr(Pm if Pm is not None else P), ptr(dS), ptr(dpooled),
                 ptr(vec), int(per_frame), stream())
        return (dvec, None, None, None, None, None, None, None, None) + dtaps


class _LSTPPoolScores(torch.autograd.Function):
    """The pooling with the raw scores [F*N, nq] supplied by the caller (SMART_LN_KEYS: scores against NORMALISED projected
    keys cannot be folded into a query-side vector): P = softmax_n(scores / sqrt(d)), pooled = sum_n P x."""

    @staticmethod
    def forward(ctx, scores, taps, F, N, T, nq, inv_sqrt_d, disjoint, holder, *grad_taps):
        ctx.want_dx = len(grad_taps) > 0 and any(t.requires_grad for t in grad_taps)
        dt = BF16 if taps[0].dtype == torch.bfloat16 else F32
        D = taps[0].shape[1]
        C = D * len(taps)
        dev = taps[0].device
        scores = scores.contiguous()
        tab = _tap_table(taps)
        P = torch.empty(F, nq, N, device=dev, dtype=torch.float32)
        Pm = torch.empty_like(P) if disjoint else None
        rowsum = torch.empty(F, nq, device=dev, dtype=torch.float32) if disjoint else None
        call('mvf_lstp_softmax_fwd', ptr(scores), ptr(P), ptr(Pm), ptr(rowsum), F, N, nq, inv_sqrt_d, int(disjoint), stream())
        pooled = torch.empty(F // T, nq, T, C, device=dev, dtype=torch.float32)
        call('mvf_lstp_wsum', tab, len(taps), dt, D, F, N, T, nq, ptr(Pm if disjoint else P), ptr(pooled), stream())
        ctx.taps = taps
        ctx.save_for_backward(P, Pm)
        ctx.cfg = (F, N, T, nq, inv_sqrt_d, dt, D, C)
        if holder is not None:
            holder['attn'] = Pm if disjoint else P
        ctx.set_materialize_grads(False)
        if disjoint:
            rs = rowsum.view(F // T, T, nq).transpose(1, 2).contiguous()
        else:
            rs = _ones((F // T, nq, T), dev).view(F // T, nq, T)
            ctx.mark_non_differentiable(rs)
        return pooled, rs

    @staticmethod
    def backward(ctx, dpooled, drs):
        P, Pm = ctx.saved_tensors
        if dpooled is None:
            dpooled = torch.zeros(ctx.cfg[0] // ctx.cfg[2], ctx.cfg[3], ctx.cfg[2], ctx.cfg[7], device=P.device, dtype=torch.float32)
        F, N, T, nq, inv_sqrt_d, dt, D, C = ctx.cfg
        taps = ctx.taps
        tab = _tap_table(taps)
        dev = dpooled.device
        dpooled = dpooled.contiguous()
        dP = torch.empty(F * N, nq, device=dev, dtype=torch.float32)
        call('mvf_lstp_scores', tab, len(taps), dt, D, F, N, T, nq, ptr(dpooled), 1, ptr(dP), stream())
        drow = None
        if Pm is not None and drs is not None:
            drow = drs.transpose(1, 2).contiguous().view(F, nq)
        dS = torch.empty(F, nq, N, device=dev, dtype=torch.float32)
        call('mvf_lstp_softmax_bwd', ptr(P), ptr(Pm), ptr(dP), ptr(drow), ptr(dS), F, N, nq, inv_sqrt_d, stream())
        dscores = dS.transpose(1, 2).reshape(F * N, nq)          # [F, nq, N] -> the caller's [F*N, nq] layout
        dtaps = ()
        if ctx.want_dx:                                           # value-side gradient only: the score side flows via `scores`
            if dt != F32:
                raise _lib.MvfError('token gradients need fp32 taps (trainable backbone blocks run in fp32)')
            dtaps = tuple(torch.empty_like(t) for t in taps)
            arr = (ctypes.c_void_p * len(taps))(*[t.data_ptr() for t in dtaps])
            call('mvf_lstp_dx', arr, len(taps), D, F, N, T, nq, ptr(Pm if Pm is not None else P), None, ptr(dpooled), None, 0,
                 stream())
        return (dscores, None, None, None, None, None, None, None, None) + dtaps


class _FrameScores(torch.autograd.Function):
    """scores[f*N + n, j] = k[f*N + n, :] . q[b, j, t, :]  (f = b*T + t): per-frame queries against explicit keys
    (SMART_LN_KEYS with dynamic queries, mvformer.py:365-400).  The pooling kernels with the keys as the only "tap":
    mvf_lstp_scores forward, mvf_lstp_wsum for dq and mvf_lstp_dx for dk."""

    @staticmethod
    def forward(ctx, k, q, F, N, T, nq):
        k, q = k.contiguous(), q.contiguous()
        d = k.shape[1]
        scores = torch.empty(F * N, nq, device=k.device, dtype=torch.float32)
        call('mvf_lstp_scores', _tap_table((k,)), 1, F32, d, F, N, T, nq, ptr(q), 1, ptr(scores), stream())
        ctx.save_for_backward(k, q)
        ctx.cfg = (F, N, T, nq, d)
        return scores

    @staticmethod
    def backward(ctx, dscores):
        k, q = ctx.saved_tensors
        F, N, T, nq, d = ctx.cfg
        dS = dscores.view(F, N, nq).transpose(1, 2).contiguous()           # the pooling kernels' [F, nq, N] layout
        dq = torch.empty(F // T, nq, T, d, device=k.device, dtype=torch.float32)
        call('mvf_lstp_wsum', _tap_table((k,)), 1, F32, d, F, N, T, nq, ptr(dS), ptr(dq), stream())
        dk = torch.empty_like(k)
        arr = (ctypes.c_void_p * 1)(dk.data_ptr())
        call('mvf_lstp_dx', arr, 1, d, F, N, T, nq, None, ptr(dS), None, ptr(q), 1, stream())     # score side only
        return dk, dq, None, None, None, None


def frame_scores(k, q, F, N, T, nq):
    """k [F*N, d] fp32 keys, q [Bc, nq, T, d] per-frame queries -> raw scores [F*N, nq]."""
    return _FrameScores.apply(k, q, F, N, T, nq)


def lstp_pool_from_scores(scores, taps, F, N, T, nq, d_model, disjoint=False, holder=None):
    taps = tuple(taps)
    grad_taps = taps if any(t.requires_grad for t in taps) else ()
    return _LSTPPoolScores.apply(scores, taps, F, N, T, nq, 1.0 / math.sqrt(d_model), disjoint, holder, *grad_taps)


def token_pool(taps, F, N, mode):
    """taps: list of [F*N, D] tensors -> [F, n_taps*D] fp32: max ('max_pool') or mean ('avg_pool') over each frame's tokens
    (late fusion).  Forward only: the tapped blocks must be frozen."""
    if any(t.requires_grad for t in taps):
        raise _lib.MvfError('late fusion over spatial tokens of TRAINABLE backbone blocks is not built (no pooling backward)')
    dt = BF16 if taps[0].dtype == torch.bfloat16 else F32
    D = taps[0].shape[1]
    out = torch.empty(F, D * len(taps), device=taps[0].device, dtype=torch.float32)
    call('mvf_token_pool', _tap_table(taps), len(taps), dt, D, F, N, {'max_pool': 0, 'avg_pool': 1}[mode], ptr(out), stream())
    return out


def lstp_pool(vec, taps, F, N, T, nq, d_model, disjoint=False, holder=None):
    """-> (pooled [Bc, nq, T, C], rowsum [Bc, nq, T]).  Taps that require grad (outputs of trainable backbone blocks) get
    their gradient from mvf_lstp_dx."""
    taps = tuple(taps)
    grad_taps = taps if any(t.requires_grad for t in taps) else ()
    return _LSTPPool.apply(vec, taps, F, N, T, nq, 1.0 / math.sqrt(d_model), disjoint, holder, *grad_taps)


# ------------------------------------------------------------------------------------------------
# SCL loss
# ------------------------------------------------------------------------------------------------
class _SCLLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, step, length, mask, T, flags, tau, var, row0, rows, grad_scale):
        emb = emb.contiguous()
        M, E = emb.shape
        dev = emb.device
        st = torch.empty(4, M, device=dev, dtype=torch.float32)  # S, R, c, lossrow
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        call('mvf_scl_fwd', ptr(emb), ptr(step), ptr(length), ptr(mask), st[0].data_ptr(), st[1].data_ptr(),
             st[2].data_ptr(), st[3].data_ptr(), ptr(loss), M, E, T, flags, tau, var, stream())
        ctx.save_for_backward(emb, step, length, mask, st)
        ctx.cfg = (T, flags, tau, var, row0, rows, grad_scale)
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        emb, step, length, mask, st = ctx.saved_tensors
        T, flags, tau, var, row0, rows, grad_scale = ctx.cfg
        M, E = emb.shape
        g = gout.reshape(1) if (grad_scale == 1.0 and gout.dtype == torch.float32) else (gout * grad_scale).reshape(1).contiguous().float()
        dE = torch.zeros(M, E, device=emb.device, dtype=torch.float32) if rows != M else \
            torch.empty(M, E, device=emb.device, dtype=torch.float32)
        call('mvf_scl_bwd', ptr(emb), ptr(step), ptr(length), ptr(mask), st[0].data_ptr(), st[1].data_ptr(),
             st[2].data_ptr(), ptr(g), dE[row0:row0 + rows].data_ptr(), M, E, T, row0, rows, flags, tau, var, stream())
        return dE, None, None, None, None, None, None, None, None, None, None


def scl_rows(steps, seq_lens, masks):
    """chosen_steps [B, V, T] int64, seq_lens [B, V] int64, video_masks [B*V, 1, T] / [B, V, T] fp32 (or None) on the device ->
    the three per-row float vectors of the loss as ONE [3, M] tensor (mvf_scl_rows), or None when the inputs are not of that
    form (the caller then converts them with tensor ops)."""
    if not (steps.is_cuda and seq_lens.is_cuda and steps.dtype == torch.int64 and seq_lens.dtype == torch.int64 and
            steps.is_contiguous() and seq_lens.is_contiguous()):
        return None
    if masks is not None and not (masks.is_cuda and masks.dtype == torch.float32 and masks.is_contiguous() and
                                  masks.numel() == steps.numel()):
        return None
    T = steps.shape[-1]
    clips = steps.numel() // T
    if seq_lens.numel() != clips:
        return None
    out = torch.empty(3, clips * T, device=steps.device, dtype=torch.float32)
    call('mvf_scl_rows', ptr(steps), ptr(seq_lens), ptr(masks), ptr(out), clips, T, stream())
    return out


def backward(loss):
    """loss.backward() with a cached seed gradient (autograd otherwise fills a fresh ones_like(loss) every step)."""
    loss.backward(_ones((), loss.device).view(()) if loss.dim() == 0 and loss.dtype == torch.float32 else None)


def scl_loss(emb, steps, seq_lens, masks, num_frames, negative_type, temperature, label_variance, row0=0, rows=None,
             grad_scale=1.0):
    """emb [M, E] rows ordered (video, view, frame); steps/seq_lens/masks per row [M].  `row0/rows`: only these
    rows receive a gradient (local slice of cross-GPU gathered embeddings)."""
    M = emb.shape[0]
    flags = (1 if 'single' in negative_type else 0) | (2 if 'noself' in negative_type else 0)
    return _SCLLoss.apply(emb, steps.reshape(-1).float().contiguous(), seq_lens.reshape(-1).float().contiguous(),
                          masks.reshape(-1).float().contiguous(), num_frames, flags, float(temperature),
                          float(label_variance), row0, M if rows is None else rows, grad_scale)


# ------------------------------------------------------------------------------------------------
# frozen ViT backbone
# ------------------------------------------------------------------------------------------------
# bf16 mode: fold LayerNorms into the GEMM that consumes them (include/mvf_hip.h: qkv_c / fc1_c).  MVF_LN_FOLD: 0 = none (the
# LayerNorm kernels), 1 = norm1 of blocks > 0 and every norm2, 2 = norm1 only, 3 = norm2 only (A/B measurements).
VIT_LN_FOLD = int(os.environ.get('MVF_LN_FOLD', '2'))
# fp8 mode: norm1 of blocks > 0 folded into the MX-fp8 qkv GEMM (the previous block's fc2 epilogue writes MX-fp8(x) + row sums).
# MVF_FP8_LN_FOLD: 1 = always, 0 = never (the layernorm_mxfp8 pass in front of every qkv GEMM), unset = where it pays: dim >= 1024.
# Same box, profiles/r06/fp8_ln_fold_ab.txt: ViT-L/14 @ 336 (dim 1024, 147 712 rows) 74.9 -> 73.5 ms per step -- the pass it removes is
# 183 us per block, the two epilogues cost 52 + 72 us; ViT-B/16 (dim 768, 50 432 rows) 8.44 -> 8.64 ms: the pass is 45 us there.
VIT_FP8_LN_FOLD = os.environ.get('MVF_FP8_LN_FOLD', '')


class PackedViT:
    """Device-resident, dtype-converted copy of a ViT's weights in the layout mvf_vit_fwd wants, plus the
    pointer tables of `struct MvfVitWeights`.  Built once per (module version, dtype): the backbone is frozen."""

    def __init__(self, sd, depth, dim, heads, patch, img, taps, dtype, ln_eps=1e-6, ln_fold=None):
        self.code, self.tdtype = _dt(dtype)
        dev = sd['cls_token'].device
        if dev.type != 'cuda':
            raise _lib.MvfError('PackedViT needs CUDA(HIP) tensors (no CPU fallback)')
        self.keep = []
        self.depth, self.dim, self.heads, self.patch, self.img = depth, dim, heads, patch, img
        self.taps = list(taps)
        # timm Attention's `q * self.scale` and the softmax's change of base folded into the frozen weights: where the attention runs on
        # the streamed kernel (mvf_vit_attn_q_prescaled: 16-bit / fp8 modes, token counts outside 193 .. 208) the q rows of every qkv
        # weight and bias carry log2(e) / 8, applied in fp64 BEFORE the one rounding / quantisation (and before a LayerNorm fold)
        tokens = (img // patch) ** 2 + 1
        self.q_prescaled = bool(depth > 0 and dim == 64 * heads and _lib.load().mvf_vit_attn_q_prescaled(self.code, tokens))
        if self.q_prescaled:
            sd = dict(sd)
            rs = torch.ones(3 * dim, dtype=torch.float64, device=dev)
            rs[:dim] = math.log2(math.e) / 8.0
            for i in range(depth):
                kw, kb = 'blocks.%d.attn.qkv.weight' % i, 'blocks.%d.attn.qkv.bias' % i
                sd[kw] = (sd[kw].detach().double() * rs[:, None]).float()
                sd[kb] = (sd[kb].detach().double() * rs).float()

        def f32(t):
            t = t.detach().float().contiguous()
            self.keep.append(t)
            return t.data_ptr()

        def mat(t):
            t = t.detach().float().contiguous()
            if self.code in (BF16, FP8):
                o = torch.empty(t.shape, device=dev, dtype=torch.bfloat16)
                call('mvf_cast_f32_bf16', t.data_ptr(), o.data_ptr(), t.numel(), stream())
                t = o
            elif self.code == F16:
                o = torch.empty(t.shape, device=dev, dtype=torch.float16)
                call('mvf_cast_f32_f16', t.data_ptr(), o.data_ptr(), t.numel(), stream())
                t = o
            self.keep.append(t)
            return t.data_ptr()

        w = _lib.MvfVitWeights()
        w.depth, w.dim, w.heads, w.patch, w.img, w.n_taps = depth, dim, heads, patch, img, len(self.taps)
        w.q_prescaled = int(self.q_prescaled)
        for i, t in enumerate(self.taps):
            w.taps[i] = t
        w.ln_eps = ln_eps
        w.cls_token = f32(sd['cls_token'].reshape(-1))
        w.pos_embed = f32(sd['pos_embed'].reshape(-1, dim))
        pw = sd['patch_embed.proj.weight'].detach().float().reshape(dim, -1)
        kp = (pw.shape[1] + 127) // 128 * 128           # K granule of the GEMM kernels (mvf_hip.h: patch_w)
        if kp != pw.shape[1]:
            pw = torch.cat([pw, pw.new_zeros(dim, kp - pw.shape[1])], 1)
        w.patch_w = mat(pw)
        w.patch_b = f32(sd['patch_embed.proj.bias'])
        w.norm_w, w.norm_b = f32(sd['norm.weight']), f32(sd['norm.bias'])

        def table(fmt, conv):
            arr = (ctypes.c_void_p * depth)(*[conv(sd[fmt % i]) for i in range(depth)])
            self.keep.append(arr)
            return ctypes.cast(arr, ctypes.POINTER(ctypes.c_void_p))
        def ptr_table(ptrs):
            arr = (ctypes.c_void_p * depth)(*ptrs)
            self.keep.append(arr)
            return ctypes.cast(arr, ctypes.POINTER(ctypes.c_void_p))

        def mx_tables(fmt, row_scale=None, col_scale=None, csum=None):
            """Frozen weights [N, K] -> MX-fp8 (e4m3 bytes + E8M0 block scales [K/128][N]) with the device quantiser, once.
            row_scale: per-layer key of a [N] vector multiplied into the rows first (LayerScale folded into proj).
            col_scale(i): a [K] vector multiplied into the columns first, or None (LayerNorm gamma folded into the consumer);
            csum: list that receives, per layer, the row sums of the DEQUANTISED result (or None where col_scale(i) is None)."""
            wp, sp = [], []
            for i in range(depth):
                W = sd[fmt % i].detach().float().contiguous()
                if row_scale is not None:
                    W = (W * sd[row_scale % i].detach().float()[:, None]).contiguous()
                cs = col_scale(i) if col_scale is not None else None
                if cs is not None:
                    W = (W.double() * cs.double()[None, :]).float().contiguous()
                n, k = W.shape
                if k % 256 != 0:
                    raise _lib.MvfError('fp8 mode needs GEMM K %% 256 == 0 (got %d)' % k)
                q = torch.empty(n, k, device=dev, dtype=torch.uint8)
                sc = torch.empty(k // 128, n, device=dev, dtype=torch.int32)
                call('mvf_quant_mxfp8', F32, W.data_ptr(), k, q.data_ptr(), k, sc.data_ptr(), n, k, stream())
                self.keep += [q, sc]
                wp.append(q.data_ptr()), sp.append(sc.data_ptr())
                if csum is not None:
                    if cs is None:
                        csum.append(None)
                    else:   # c[n] = sum_k of the values the matrix cores multiply: e4m3 byte x 2^(scale byte - 127), block by block
                        e = sc.view(torch.uint8).reshape(k // 128, n, 4).permute(1, 0, 2).reshape(n, k // 32).double() - 127.0
                        deq = q.view(torch.float8_e4m3fn).double().reshape(n, k // 32, 32) * torch.exp2(e)[:, :, None]
                        csum.append(deq.sum((1, 2)))
            return ptr_table(wp), ptr_table(sp)

        def folded(lin, norm, skip0):
            """LayerNorm `norm` folded into the Linear `lin` that consumes it: per block (W' = gamma (.) W in bf16,
            d = b + W beta, c[n] = sum_k bf16(W')[n, k]) -- c is taken from the ROUNDED weights the matrix cores multiply, so
            that the row-mean term cancels exactly.  Block 0's norm1 (skip0) keeps the LayerNorm kernel."""
            wp, bp, cp = [], [], []
            for i in range(depth):
                W = sd['blocks.%d.%s.weight' % (i, lin)].detach().double()
                bias = sd['blocks.%d.%s.bias' % (i, lin)].detach().double()
                if skip0 and i == 0:
                    wp.append(mat(W.float())), bp.append(f32(bias.float())), cp.append(None)
                    continue
                g = sd['blocks.%d.%s.weight' % (i, norm)].detach().double()
                beta = sd['blocks.%d.%s.bias' % (i, norm)].detach().double()
                wp.append(mat((W * g[None, :]).float()))
                c = self.keep[-1].double().sum(1)                   # the bf16 tensor `mat` just stored
                cp.append(f32(c.float())), bp.append(f32((bias + W @ beta).float()))
            return ptr_table(wp), ptr_table(bp), ptr_table(cp)

        fold = int(VIT_LN_FOLD if ln_fold is None else ln_fold) if (self.code in (BF16, F16) and dim % 128 == 0 and depth > 0) else 0
        # fp8: norm1 of blocks > 0 only (what ln_fold = 2 means in the 16-bit modes; any non-zero ln_fold asks for it)
        want8 = (int(VIT_FP8_LN_FOLD) != 0 if VIT_FP8_LN_FOLD != '' else dim >= 1024) if ln_fold is None else int(ln_fold) != 0
        fold8 = self.code == FP8 and depth > 1 and want8
        self.ln_fold = 2 if fold8 else fold
        # Deferred residual of the attention branch (csrc/vit_fwd.hip) for LayerScale models: x + gamma_1 (.) (o W^T + b) =
        # x + o (gamma_1 (.) W)^T + gamma_1 (.) b -- gamma_1 goes into proj's weights and bias here and ls1 stays NULL, so the
        # device takes the deferred path (plain-store proj, branch output added in LayerNorm 2 and the fc2 epilogue)
        has_ls = 'blocks.0.ls1.gamma' in sd
        self.ls1_folded = bool(has_ls and depth > 0 and self.code in (BF16, FP8, F16) and dim % 128 == 0 and fold not in (1, 3)
                               and os.environ.get('MVF_PROJ_DEFER', '1') != '0')

        def proj_w(t, i):
            return t * sd['blocks.%d.ls1.gamma' % i].detach().to(t.dtype)[:, None] if self.ls1_folded else t

        def proj_b(t, i):
            return t * sd['blocks.%d.ls1.gamma' % i].detach().to(t.dtype) if self.ls1_folded else t
        w.ln1_w, w.ln1_b = table('blocks.%d.norm1.weight', f32), table('blocks.%d.norm1.bias', f32)
        w.proj_b = ptr_table([f32(proj_b(sd['blocks.%d.attn.proj.bias' % i].detach().float(), i)) for i in range(depth)])
        w.fc2_b = table('blocks.%d.mlp.fc2.bias', f32)
        w.ln2_w, w.ln2_b = table('blocks.%d.norm2.weight', f32), table('blocks.%d.norm2.bias', f32)
        if self.code != FP8:
            w.proj_w = ptr_table([mat(proj_w(sd['blocks.%d.attn.proj.weight' % i].detach().float(), i)) for i in range(depth)])
            w.fc2_w = table('blocks.%d.mlp.fc2.weight', mat)
        if self.code == FP8:
            w.proj_w, w.proj_s = mx_tables('blocks.%d.attn.proj.weight', 'blocks.%d.ls1.gamma' if self.ls1_folded else None)
            w.fc2_w, w.fc2_s = mx_tables('blocks.%d.mlp.fc2.weight')
            w.fc1_w, w.fc1_s = mx_tables('blocks.%d.mlp.fc1.weight')
            w.fc1_b = table('blocks.%d.mlp.fc1.bias', f32)
            if fold8:
                # W' = MX-fp8(gamma (.) W), d = b + W beta, c = row sums of the dequantised W' (the row-mean term then cancels exactly
                # against the products the matrix cores form); block 0's norm1 follows the patch embedding and keeps its LayerNorm pass
                csum = []
                w.qkv_w, w.qkv_s = mx_tables('blocks.%d.attn.qkv.weight', csum=csum,
                                             col_scale=lambda i: None if i == 0 else sd['blocks.%d.norm1.weight' % i].detach())
                bp, cp = [], []
                for i in range(depth):
                    bias = sd['blocks.%d.attn.qkv.bias' % i].detach().double()
                    if i == 0:
                        bp.append(f32(bias.float())), cp.append(None)
                        continue
                    W = sd['blocks.%d.attn.qkv.weight' % i].detach().double()
                    bp.append(f32((bias + W @ sd['blocks.%d.norm1.bias' % i].detach().double()).float()))
                    cp.append(f32(csum[i].float()))
                w.qkv_b, w.qkv_c = ptr_table(bp), ptr_table(cp)
            else:
                w.qkv_w, w.qkv_s = mx_tables('blocks.%d.attn.qkv.weight')
                w.qkv_b = table('blocks.%d.attn.qkv.bias', f32)
        else:
            if fold in (1, 2):
                w.qkv_w, w.qkv_b, w.qkv_c = folded('attn.qkv', 'norm1', skip0=True)
            else:
                w.qkv_w, w.qkv_b = table('blocks.%d.attn.qkv.weight', mat), table('blocks.%d.attn.qkv.bias', f32)
            if fold in (1, 3):
                w.fc1_w, w.fc1_b, w.fc1_c = folded('mlp.fc1', 'norm2', skip0=False)
            else:
                w.fc1_w, w.fc1_b = table('blocks.%d.mlp.fc1.weight', mat), table('blocks.%d.mlp.fc1.bias', f32)
        if has_ls:
            if not self.ls1_folded:
                w.ls1 = table('blocks.%d.ls1.gamma', f32)
            w.ls2 = table('blocks.%d.ls2.gamma', f32)
        self.struct = w
        self.ws = {}
        self.ws_key = None

    def workspace(self, fc, device, slot=0):
        """One activation workspace per concurrent forward (`slot`), reused call after call on that slot's stream."""
        tokens = (self.img // self.patch) ** 2 + 1
        key = (fc, tokens)
        if self.ws_key != key:
            self.ws, self.ws_key = {}, key
        if slot not in self.ws:
            nbytes = _lib.load().mvf_vit_workspace_bytes(self.code, fc, tokens, self.dim, self.patch)
            self.ws[slot] = torch.empty(nbytes, device=device, dtype=torch.uint8)
        return self.ws[slot]

    def lane_stream(self, slot, device):
        """Extra HIP stream of concurrent-forward lane `slot` >= 1 (lane 0 is the caller's stream)."""
        return backbone_stream('lane%d' % slot, device)


# Rows (frames x tokens) per lane from which a forward is split into concurrent lanes: every GEMM of a lane must still
# cover the chip (> 256 tiles of 256 x 256 for N = 768) for the split to pay.
VIT_LANE_MIN_ROWS = 22000
# HIP priority of the streams the frozen backbone runs on (lane streams here, the lookahead stream of the model) and the
# lane count of a split forward.  Measured in the full step (B = 4, 32 frames): priority -1 = priority 0 (12.14 vs 12.13 ms),
# 4 lanes 12.75 ms, 2 lanes 12.13 ms, 1 lane 12.48 ms.
BACKBONE_STREAM_PRIORITY = int(os.environ.get('MVF_BACKBONE_PRIORITY', '0'))      # (the variable: A/B measurements)
VIT_LANES = 2


_BACKBONE_STREAMS = {}


def backbone_stream(role, device):
    """The process's HIP stream for a backbone role on a device ('side' = the lookahead stream of models/transformer.py, 'lane1', ..):
    ONE per role and device for every model of the process.  HIP deals streams over four hardware queues in creation order; a second
    model that created its own (an evaluation copy, a sweep) got queues that the first model's streams or the caller's stream already
    used, and two streams on one hardware queue run one after the other -- measured: the same backbone forward 10.5 ms in a fresh process,
    12.5 ms as the second model of a process (tools/order_probe.py)."""
    device = torch.device(device)
    key = (role, device.index if device.index is not None else torch.cuda.current_device())
    st = _BACKBONE_STREAMS.get(key)
    if st is None:
        st = _BACKBONE_STREAMS[key] = torch.cuda.Stream(device=device, priority=BACKBONE_STREAM_PRIORITY)
    return st


def vit_forward(frames, packed, frames_per_chunk=0, want_cls=True, attn_variant=0, lanes=None):
    """frames [F,3,H,W] fp32 -> (taps: list of [F*(N-1), dim] tensors in packed.tdtype, cls [F, dim] fp32 | None).

    `lanes`: the F frames are forwarded as `lanes` independent slices (of F // lanes frames, the first F % lanes one more) on as
    many HIP streams (lane 0 = the caller's stream), each with its own workspace and writing its slice of the outputs.  The persistent GEMM leaves most CUs idle
    in the last, partially filled round of 256 x 256 tiles (N = 768: 591 tiles on 256 CUs); with two forwards in flight
    the other lane's kernels fill those CUs (ViT-B/16, 256 frames: 12.09 -> 11.31 ms, outputs bitwise identical).
    None = 2 lanes when each still fills the chip, else 1."""
    if not frames.is_cuda:
        raise _lib.MvfError('vit_forward received a %s tensor (no CPU fallback)' % frames.device)
    frames = frames.contiguous().float()
    F = frames.shape[0]
    assert frames.shape[1:] == (3, packed.img, packed.img), frames.shape
    np_ = (packed.img // packed.patch) ** 2
    if lanes is None:
        lanes = VIT_LANES if (F % VIT_LANES == 0 and (F // VIT_LANES) * (np_ + 1) >= VIT_LANE_MIN_ROWS) else 1
    if lanes < 1 or lanes > F:
        raise _lib.MvfError('vit_forward: %d frames do not split into %d lanes' % (F, lanes))
    sizes = [F // lanes + (1 if s < F % lanes else 0) for s in range(lanes)]       # (the first F % lanes lanes take one frame more)
    starts = [sum(sizes[:s]) for s in range(lanes)]
    fc_max = sizes[0] if frames_per_chunk <= 0 else min(frames_per_chunk, sizes[0])
    dev = frames.device
    taps = [torch.empty(F * np_, packed.dim, device=dev, dtype=packed.tdtype) for _ in packed.taps]
    cls = torch.empty(F, packed.dim, device=dev, dtype=torch.float32) if want_cls else None
    esz = taps[0].element_size() if taps else 0
    cur = torch.cuda.current_stream(dev)
    ready = None
    if lanes > 1:
        ready = torch.cuda.Event()
        ready.record(cur)                      # inputs written / outputs allocated in caller-stream order
    joins = []
    for s in range(lanes):
        st = cur if s == 0 else packed.lane_stream(s, dev)
        if s > 0:
            st.wait_event(ready)
        fl, f0 = sizes[s], starts[s]
        fc = fl if frames_per_chunk <= 0 else min(frames_per_chunk, fl)
        ws = packed.workspace(fc_max, dev, s)      # (one size for every lane: the cache is keyed by it)
        tab = (ctypes.c_void_p * max(len(taps), 1))(*[t.data_ptr() + f0 * np_ * packed.dim * esz for t in taps])
        call('mvf_vit_fwd', ctypes.byref(packed.struct), packed.code,
             frames.data_ptr() + f0 * 3 * packed.img * packed.img * 4, fl, tab,
             None if cls is None else cls.data_ptr() + f0 * packed.dim * 4, ptr(ws), ws.numel(), fc, attn_variant,
             ctypes.c_void_p(st.cuda_stream))
        if s > 0:
            ev = torch.cuda.Event()
            ev.record(st)
            joins.append(ev)
            for t in taps + [frames] + ([cls] if cls is not None else []):
                t.record_stream(st)
    for ev in joins:
        cur.wait_event(ev)
    return taps, cls


# ------------------------------------------------------------------------------------------------
# view augmentation
# ------------------------------------------------------------------------------------------------
_aug_ws = {}


def augment_clips(clips, params, size):
    """clips [n, T, 3, H, W] fp32 in [0, 1] on the device, params: n `_lib.MvfAugmentParams` (one clip's draws each)
    -> [n, T, 3, size, size] fp32, normalised (include/mvf_hip.h: mvf_augment_clips)."""
    if not clips.is_cuda:
        raise _lib.MvfError('augment_clips received a %s tensor (no CPU fallback)' % clips.device)
    clips = clips.contiguous().float()
    n, T, c, H, W = clips.shape
    assert c == 3 and len(params) == n, (clips.shape, len(params))
    nbytes = _lib.load().mvf_augment_workspace_bytes(n, T, size)
    key = (clips.device.index, torch.cuda.current_stream(clips.device).cuda_stream)
    ws = _aug_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _aug_ws[key] = torch.empty(nbytes, device=clips.device, dtype=torch.uint8)
    out = torch.empty(n, T, 3, size, size, device=clips.device, dtype=torch.float32)
    arr = (_lib.MvfAugmentParams * n)(*params)
    call('mvf_augment_clips', clips.data_ptr(), out.data_ptr(), n, T, H, W, size, ctypes.cast(arr, ctypes.c_void_p),
         ws.data_ptr(), ws.numel(), stream())
    return out


def vit_front(frames, packed, frames_per_chunk=0):
    """frames [F,3,H,W] fp32 -> fp32 residual stream [F, N, dim] (CLS row included) after the packed.depth frozen blocks:
    the front end of a partially frozen backbone (mvf_vit_fwd_x; depth 0 = patch + position embedding only)."""
    if not frames.is_cuda:
        raise _lib.MvfError('vit_front received a %s tensor (no CPU fallback)' % frames.device)
    frames = frames.contiguous().float()
    F = frames.shape[0]
    n = (packed.img // packed.patch) ** 2 + 1
    fc = F if frames_per_chunk <= 0 else min(frames_per_chunk, F)
    ws = packed.workspace(fc, frames.device)
    x = torch.empty(F, n, packed.dim, device=frames.device, dtype=torch.float32)
    call('mvf_vit_fwd_x', ctypes.byref(packed.struct), packed.code, ptr(frames), F, None, None, ptr(x), ptr(ws), ws.numel(),
         fc, 0, stream())
    return x


def vit_blocks(x, packed, first_block, n_blocks=1):
    """x [F, N, dim] fp32 residual stream -> the stream after blocks [first_block, first_block + n_blocks) (a new tensor):
    mvf_vit_blocks_fwd, for the per-block parity checks (every block fed the oracle's input)."""
    if not x.is_cuda:
        raise _lib.MvfError('vit_blocks received a %s tensor (no CPU fallback)' % x.device)
    x = x.float().contiguous().clone()
    F = x.shape[0]
    ws = packed.workspace(F, x.device)
    call('mvf_vit_blocks_fwd', ctypes.byref(packed.struct), packed.code, ptr(x), F, first_block, n_blocks, ptr(ws), ws.numel(),
         0, stream())
    return x


# ------------------------------------------------------------------------------------------------
# optimiser
# ------------------------------------------------------------------------------------------------
def grad_norm(flat_grad, scratch, out, extra_sq=None):
    call('mvf_grad_norm', ptr(flat_grad), flat_grad.numel(), ptr(extra_sq), ptr(scratch), ptr(out), stream())
    return out


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, clip=0.0, norm=None, gscale=1.0, zero_grad=False):
    """zero_grad: the kernel zeroes g as it consumes it (utils.optimizer.FusedAdam: the next zero_grad() becomes free)."""
    call('mvf_adam_step', ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step, clip,
         ptr(norm), gscale, int(zero_grad), stream())
