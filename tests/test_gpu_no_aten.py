"""The steady-state training step launches only the library's own kernels: no at::native elementwise /
reduce / fill kernels and no runtime copy kernels from autograd's gradient accumulation, broadcast glue or dtype casts.
Plus the pieces that made it so, each against its plain composition."""
import pytest
import torch
from torch.profiler import profile, ProfilerActivity

pytestmark = pytest.mark.gpu

from video_rep_learning_amd import ops  # noqa: E402
from video_rep_learning_amd.algos import get_algo  # noqa: E402
from video_rep_learning_amd.train import DataParallelModel  # noqa: E402
from video_rep_learning_amd.utils.optimizer import construct_optimizer  # noqa: E402
import test_gpu_model as T  # noqa: E402

DEV = 'cuda'


def _device_kernels(fn, steps):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
    names = []
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            names.append(ev.name)
    return names


@pytest.mark.parametrize('padded', [False, True])
def test_training_step_launches_no_aten_kernels(padded):
    cfg, model = T.make(3, **dict(T.SMALL, dropout=0.1, compute_dtype='bf16', batch_size=2))
    wrapped = DataParallelModel(model)
    opt = construct_optimizer(wrapped, cfg)
    algo = get_algo(cfg)
    videos, seq_lens, steps, masks = [t.to(DEV) for t in T.batch(cfg, 4, pad=2 if padded else 0)]
    model.train()

    def step():
        wrapped.prefetch(videos)
        opt.zero_grad()
        loss = algo.compute_loss(wrapped, videos, seq_lens, steps, masks)['loss']
        ops.backward(loss)
        opt.step(max_norm=cfg.OPTIMIZER.GRAD_CLIP)

    wrapped.prefetch(videos)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    names = _device_kernels(step, 2)
    foreign = sorted({n for n in names if 'at::native' in n or 'rocclr' in n or 'Memcpy' in n or 'Memset' in n})
    assert names and not foreign, foreign


def _forced_reducer_worker(rank, port, ret):
    """One process, one GPU, a ONE-rank `nccl` (= RCCL) group with MVF_FORCE_REDUCER=1: every collective call site of the data-parallel
    step runs (bucketed async gradient all-reduce, SyncBatchNorm all-gather + merge, backward column-sum all-reduce)."""
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', MVF_FORCE_REDUCER='1')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        from video_rep_learning_amd.utils import distributed as du
        assert du.collectives_active()
        cfg, model = T.make(3, **dict(T.SMALL, dropout=0.1, compute_dtype='bf16', batch_size=2, head_dtype=ret['head']))
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
        wrapped = DataParallelModel(model)
        opt = construct_optimizer(wrapped, cfg)
        algo = get_algo(cfg)
        videos, seq_lens, steps, masks = [t.to(DEV) for t in T.batch(cfg, 4)]
        model.train()

        def step():
            wrapped.prefetch(videos)
            opt.zero_grad()
            loss = algo.compute_loss(wrapped, videos, seq_lens, steps, masks)['loss']
            ops.backward(loss)
            opt.step(max_norm=cfg.OPTIMIZER.GRAD_CLIP)

        wrapped.prefetch(videos)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        names = _device_kernels(step, 2)
        ret['names'] = sorted(set(names))
        ret['reducer_active'] = bool(opt.reducer is not None and opt.reducer.active)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('head', ['bf16', 'fp32'])
def test_data_parallel_step_launches_no_aten_kernels_either(head):
    """The same census with every collective of the data-parallel step live: besides the library's kernels only
    RCCL's own work may appear (its kernels, and the device-to-device copy a one-rank all-gather degenerates to) -- no at::native
    kernel from assembling / merging the SyncBatchNorm statistics (mvf_syncbn_merge + a persistent exchange block replaced cat /
    stack / sum / pow ...), no fill, no host copies."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    ret['head'] = head
    mp.spawn(_forced_reducer_worker, args=(port, ret), nprocs=1, join=True)
    names = list(ret['names'])
    assert ret['reducer_active'] and names
    foreign = [n for n in names if 'at::native' in n or 'Memset' in n or 'HtoD' in n or 'DtoH' in n or
               (('Memcpy' in n or 'rocclr' in n) and 'DtoD' not in n)]
    assert not foreign, foreign
    assert any('syncbn_merge' in n for n in names), names


def test_layer_norm_fork_equals_layer_norm_plus_residual_gradient():
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(6, 24, 256, generator=g).to(DEV)
    w = (1 + 0.1 * torch.randn(256, generator=g)).to(DEV).requires_grad_(True)
    b = (0.1 * torch.randn(256, generator=g)).to(DEV).requires_grad_(True)
    dy = torch.randn(6, 24, 256, generator=g).to(DEV)
    outs = []
    for fork in (False, True):
        x = x0.clone().requires_grad_(True)
        w.grad = b.grad = None
        if fork:
            xr, h = ops.layer_norm_fork(x, w, b, 1e-5)
        else:
            xr, h = x, ops.layer_norm(x, w, b, 1e-5)
        y = xr + 2.0 * h * h
        y.backward(dy)
        outs.append((y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone()))
    for a_, b_ in zip(*outs):
        assert torch.allclose(a_, b_, rtol=1e-5, atol=1e-5), (a_ - b_).abs().max()
    assert torch.equal(outs[0][0], outs[1][0])


def test_periodic_mask_equals_the_tiled_mask_bitwise():
    g = torch.Generator().manual_seed(1)
    B, ntok, Tn, H, Dm = 4, 3, 8, 8, 256
    S = ntok * Tn
    qkv0 = torch.randn(B * S, 3 * Dm, generator=g).to(DEV)
    mask = (torch.rand(B, Tn, generator=g) > 0.3).float().to(DEV)
    mask[:, 0] = 1.0
    d_o = torch.randn(B * S, Dm, generator=g).to(DEV)
    res = []
    for m in (mask, mask.view(B, 1, Tn).expand(B, ntok, Tn).reshape(B, S).contiguous()):
        qkv = qkv0.clone().requires_grad_(True)
        o = ops.temporal_attention(qkv, m, B, S, H)
        o.backward(d_o)
        res.append((o.detach().clone(), qkv.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_scl_rows_equals_the_tensor_ops():
    g = torch.Generator().manual_seed(2)
    B, V, Tn = 3, 2, 8
    steps = torch.sort(torch.randint(0, 100, (B, V, Tn), generator=g), dim=-1)[0].to(DEV)
    lens = torch.randint(20, 100, (B, V), generator=g).to(DEV)
    masks = (torch.rand(B * V, 1, Tn, generator=g) > 0.2).float().to(DEV)
    rows = ops.scl_rows(steps, lens, masks)
    M = B * V * Tn
    assert torch.equal(rows[0], steps.reshape(M).float())
    assert torch.equal(rows[1], lens.reshape(B, V, 1).expand(B, V, Tn).reshape(M).float())
    assert torch.equal(rows[2], masks.reshape(M))
    assert ops.scl_rows(steps.float(), lens, masks) is None        # other dtypes: the caller's tensor ops


def test_static_query_gradients_land_in_the_flat_buffers():
    """(Q_s + Q_s_b) W_K through ops.static_query (in-slot gradients) against the autograd composition, via a whole small model:
    one backward with the fused optimizer's flat buffers, one on a plain copy."""
    cfg, model = T.make(5, **T.SMALL)
    cfg2, ref = T.make(5, **T.SMALL)
    ref.load_state_dict(model.state_dict())
    wrapped = DataParallelModel(model)
    opt = construct_optimizer(wrapped, cfg)        # re-homes the parameters: static_query takes the slot path
    videos, seq_lens, steps, masks = [t.to(DEV) for t in T.batch(cfg, 6)]
    model.train(), ref.train()
    opt.zero_grad()
    ops.backward(get_algo(cfg).compute_loss(wrapped, videos, seq_lens, steps, masks)['loss'])
    get_algo(cfg2).compute_loss(ref, videos, seq_lens, steps, masks)['loss'].backward()
    torch.cuda.synchronize()
    ca, cb = model.embed.pooling.cross_att, ref.embed.pooling.cross_att
    for name, a_, b_ in (('Q_s', ca.Q_s.grad, cb.Q_s.grad), ('Q_s_b', ca.Q_s_b.grad, cb.Q_s_b.grad),
                         ('linear_K2d.weight', ca.linear_K2d.weight.grad, cb.linear_K2d.weight.grad)):
        e = ((a_ - b_).abs().max() / b_.abs().max()).item()        # fp32 summation order only
        assert e <= 5e-5, (name, e)


def test_zero_grad_after_step_is_free_and_still_correct_in_odd_orders():
    cfg, model = T.make(7, **T.SMALL)
    wrapped = DataParallelModel(model)
    opt = construct_optimizer(wrapped, cfg)
    algo = get_algo(cfg)
    videos, seq_lens, steps, masks = [t.to(DEV) for t in T.batch(cfg, 8)]
    model.train()
    flat = opt.flat
    opt.zero_grad()
    ops.backward(algo.compute_loss(wrapped, videos, seq_lens, steps, masks)['loss'])
    assert flat.dirty and flat.flat_g.abs().sum().item() > 0
    g1 = flat.flat_g.clone()
    opt.step(max_norm=10.0)
    assert not flat.dirty and flat.flat_g.abs().sum().item() == 0.0      # the Adam kernel zeroed what it consumed
    opt.zero_grad()                                                         # nothing to do
    # odd order: a backward WITHOUT a step in between, then zero_grad must really zero
    ops.backward(algo.compute_loss(wrapped, videos, seq_lens, steps, masks)['loss'])
    assert flat.dirty and flat.flat_g.abs().sum().item() > 0
    opt.zero_grad()
    assert flat.flat_g.abs().sum().item() == 0.0
    # and two backwards accumulate
    ops.backward(algo.compute_loss(wrapped, videos, seq_lens, steps, masks)['loss'])
    ga = flat.flat_g.clone()
    ops.backward(algo.compute_loss(wrapped, videos, seq_lens, steps, masks)['loss'])
    assert torch.allclose(flat.flat_g, 2 * ga, rtol=1e-4, atol=1e-7)
    assert g1.shape == ga.shape
