"""Multi-GPU host logic on CPU: world_size-2 `gloo` process groups (spawned here) exercise every collective of the
data-parallel path that does not need a kernel -- the flat-buffer gradient bucketing + async all-reduce (C1), the
SyncBN statistics merge (C2), the autograd-aware embedding gather (C9) against the oracle's loss on rank-concatenated
inputs, and the loss all-reduce helper (C4)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from video_rep_learning_amd.utils import distributed as du
from oracle import scl as OS


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def spawn(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


# ---------------------------------------------------------------------------------------------- C1
def _grad_reducer(rank, world):
    torch.manual_seed(0)
    lin1, lin2 = torch.nn.Linear(8, 16), torch.nn.Linear(16, 4)
    params = list(lin1.parameters()) + list(lin2.parameters())
    flat = du.FlatBuffers(params)
    red = du.GradReducer(flat, bucket_bytes=256)          # several buckets
    assert len(red.buckets) > 1
    g = torch.Generator().manual_seed(100)
    xs = torch.randn(world, 5, 8, generator=g)
    flat.zero_grad()
    loss = lin2(torch.relu(lin1(xs[rank]))).pow(2).sum()
    loss.backward()                                        # hooks launch bucket all-reduces as grads become ready
    scale = red.finish()
    got = flat.flat_g.clone() * scale
    # single-process reference: average of the per-rank gradients
    ref = torch.zeros_like(got)
    for r in range(world):
        for p in params:
            p.grad = None
        l = lin2(torch.relu(lin1(xs[r]))).pow(2).sum()
        gs = torch.autograd.grad(l, params)
        for p, o, gg in zip(params, flat.offsets, gs):
            ref[o:o + p.numel()] += gg.reshape(-1) / world
    return (got - ref).abs().max().item(), scale


def test_flat_gradient_allreduce_equals_mean_of_rank_gradients():
    for err, scale in spawn(_grad_reducer):
        assert err < 1e-6 and scale == 0.5


# ---------------------------------------------------------------------------------------------- C2
def _sync_bn(rank, world):
    from video_rep_learning_amd import ops
    g = torch.Generator().manual_seed(7)
    rows = [5, 9]                                           # ragged: different row counts per rank
    x = [torch.randn(n, 6, generator=g) * (r + 1) + r for r, n in enumerate(rows)]
    mine = x[rank]
    mean, var, count = ops.sync_bn_stats(mine.mean(0), mine.var(0, unbiased=False), mine.shape[0])
    full = torch.cat(x)
    return ((mean - full.mean(0)).abs().max().item(), (var - full.var(0, unbiased=False)).abs().max().item(), count)


def test_syncbn_statistics_equal_concatenated_batch():
    for e_mean, e_var, count in spawn(_sync_bn):
        assert e_mean < 1e-6 and e_var < 1e-5 and count == 14.0


# ---------------------------------------------------------------------------------------------- C9
def _gathered_scl(rank, world):
    g = torch.Generator().manual_seed(3)
    b, t, e = 2, 6, 8
    emb = torch.nn.functional.normalize(torch.randn(world, b, 2, t, e, generator=g), dim=-1)
    steps = torch.sort(torch.randint(0, 40, (world, b, 2, t), generator=g), dim=-1)[0]
    lens = torch.full((world, b, 2), 40)
    masks = torch.ones(world, b * 2, 1, t)
    kw = dict(negative_type='batch_noself', temperature=0.1, label_variance=10.0)
    # every rank: gather rows (autograd-aware), evaluate the GLOBAL loss with the oracle, backprop to its local rows
    local = emb[rank].clone().requires_grad_(True)
    rows = du.gather_rows(local.reshape(b * 2 * t, e))
    st, ln, mk = du.all_gather([steps[rank].reshape(-1), lens[rank].reshape(-1), masks[rank].reshape(-1)])
    loss = OS.scl_loss(rows.reshape(world * b, 2, t, e), ln.reshape(world * b, 2), st.reshape(world * b, 2, t),
                       mk.reshape(world * b * 2, 1, t), **kw)
    loss.backward()
    # single-process reference on the rank-concatenated batch
    allemb = emb.reshape(world * b, 2, t, e).clone().requires_grad_(True)
    ref = OS.scl_loss(allemb, lens.reshape(world * b, 2), steps.reshape(world * b, 2, t),
                      masks.reshape(world * b * 2, 1, t), **kw)
    ref.backward()
    gref = allemb.grad.reshape(world, b, 2, t, e)[rank]
    return abs(loss.item() - ref.item()), (local.grad - gref).abs().max().item()


def test_gathered_scl_loss_and_local_gradient_equal_concatenated_reference():
    for e_loss, e_grad in spawn(_gathered_scl):
        assert e_loss < 1e-6 and e_grad < 1e-6


class _OracleSclOp(torch.autograd.Function):
    """Stands in for ops.scl_loss (a HIP kernel pair) with ITS contract on the CPU: per-row float vectors in, the loss over ALL rows
    out, and in the backward only rows [row0, row0 + rows) of dE, multiplied by grad_scale."""

    @staticmethod
    def forward(ctx, e, st, ln, mk, t, neg, tau, var, row0, rows, grad_scale):
        m, c = e.shape
        b = m // (2 * t)
        with torch.enable_grad():
            leaf = e.detach().clone().requires_grad_(True)
            loss = OS.scl_loss(leaf.reshape(b, 2, t, c), ln.reshape(b, 2, t)[:, :, 0].long(), st.reshape(b, 2, t).long(),
                               mk.reshape(b * 2, 1, t), negative_type=neg, temperature=tau, label_variance=var)
            (g,) = torch.autograd.grad(loss, leaf)
        rows = m if rows is None else rows
        d = torch.zeros_like(g)
        d[row0:row0 + rows] = g[row0:row0 + rows] * grad_scale
        ctx.save_for_backward(d)
        return loss.detach()

    @staticmethod
    def backward(ctx, gout):
        (d,) = ctx.saved_tensors
        return (d * gout,) + (None,) * 10


def _gather_mode_parameter_gradient(rank, world):
    """The PRODUCT's SCL.compute_sequence_loss in gather mode (algos/scl.py: gather_rows + grad_scale = W) behind a small trainable
    map, gradients averaged by the flat-buffer reducer as in training: the PARAMETER gradient must equal the single-process gradient
    of the loss on the rank-concatenated batch.  (An Adam-update comparison cannot see a wrong scale -- Adam is scale invariant.)"""
    from video_rep_learning_amd import ops
    from video_rep_learning_amd.algos.scl import SCL
    from video_rep_learning_amd.utils import presets
    cfg = presets.make_cfg(num_frames=6, batch_size=2)
    cfg.SCL.NEGATIVE_TYPE = 'batch_noself'
    cfg.MI355X = {'GATHER_EMBEDDINGS': True}
    algo = SCL(cfg)
    assert algo.gather and du.collectives_active()
    real = ops.scl_loss
    ops.scl_loss = lambda e, st, ln, mk, t, neg, tau, var, row0=0, rows=None, grad_scale=1.0: \
        _OracleSclOp.apply(e, st, ln, mk, t, neg, tau, var, row0, rows, grad_scale)
    try:
        g = torch.Generator().manual_seed(5)
        b, t, cin, e = 2, 6, 5, 8
        x = torch.randn(world, b, 2, t, cin, generator=g)
        steps = torch.sort(torch.randint(0, 40, (world, b, 2, t), generator=g), dim=-1)[0]
        lens = torch.full((world, b, 2), 40)
        masks = torch.ones(world, b * 2, 1, t)
        torch.manual_seed(7)
        lin = torch.nn.Linear(cin, e, bias=False)
        w0 = lin.weight.detach().clone()
        flat = du.FlatBuffers(list(lin.parameters()))
        red = du.GradReducer(flat, bucket_bytes=1 << 20)
        flat.zero_grad()
        emb = torch.nn.functional.normalize(lin(x[rank]), dim=-1)
        loss = algo.compute_sequence_loss(emb, lens[rank], steps[rank], masks[rank])['loss']
        loss.backward()
        scale = red.finish()
        got = (flat.flat_g * scale)[:w0.numel()].view_as(w0).clone()
    finally:
        ops.scl_loss = real
    # single process, rank-concatenated batch
    wr = w0.clone().requires_grad_(True)
    embr = torch.nn.functional.normalize(x.reshape(world * b, 2, t, cin) @ wr.t(), dim=-1)
    ref = OS.scl_loss(embr, lens.reshape(world * b, 2), steps.reshape(world * b, 2, t), masks.reshape(world * b * 2, 1, t),
                      negative_type='batch_noself', temperature=cfg.SCL.SOFTMAX_TEMPERATURE, label_variance=cfg.SCL.LABEL_VARIENCE)
    (gref,) = torch.autograd.grad(ref, wr)
    return abs(loss.item() - ref.item()), ((got - gref).abs().max() / gref.abs().max()).item(), (got.norm() / gref.norm()).item()


def test_gather_mode_parameter_gradient_equals_concatenated_reference():
    for e_loss, e_grad, ratio in spawn(_gather_mode_parameter_gradient):
        assert e_loss < 1e-6 and e_grad < 1e-5, (e_loss, e_grad, ratio)
        assert abs(ratio - 1.0) < 1e-5          # a missing x W (or a double one) shows up here as 1 / W (or W)


# ---------------------------------------------------------------------------------------------- C4
def _loss_allreduce(rank, world):
    v = torch.tensor(float(rank + 1))
    return du.all_reduce([v])[0].item(), du.get_world_size(), du.is_root_proc()


def test_loss_allreduce_average():
    out = spawn(_loss_allreduce)
    assert [o[0] for o in out] == [1.5, 1.5] and out[0][1] == 2 and out[0][2] and not out[1][2]
