"""Data-parallel helpers: one process per GPU over torch.distributed ('nccl' = RCCL over xGMI on ROCm, 'gloo'
for CPU plumbing tests).

Mirrors the helpers of CARL_MVF/utils/distributed.py that the train path calls (all_reduce :38-54,
synchronize :136-148, rank/size helpers) and replaces `DistributedDataParallel(find_unused_parameters=True)`
of train.py:285-286 -- which registers the 86 M frozen backbone parameters too -- by `GradReducer`:
gradients of the 4.8 M TRAINABLE parameters live in ONE flat fp32 buffer cut into a few buckets; a bucket's
all-reduce is launched asynchronously (RCCL side stream) from the autograd hook of its last gradient, so it
overlaps the rest of backward; the optimizer waits, then divides by world size inside the fused Adam kernel.
Also: the autograd-aware embedding all-gather that enlarges the SCL negative set (new)."""
import os

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def collectives_active():
    """True when the data-path collectives must be issued: more than one rank, or a ONE-rank process group with
    MVF_FORCE_REDUCER=1.  The forced form drives every collective call site of the step (bucketed async gradient all-reduce
    from the hooks, SyncBN statistics all-gather / all-reduce, embedding all-gather, loss all-reduce) through the backend --
    RCCL on a single-GPU box -- where they are arithmetic identities: the step must come out bitwise equal to the plain one
    (tests/test_gpu_ddp.py)."""
    return is_dist() and (dist.get_world_size() > 1 or os.environ.get('MVF_FORCE_REDUCER', '0') == '1')


def reserve_collective_cus(device=None):
    """Data-parallel runs: keep `MVF_RCCL_CUS` (default 8 = one per XCD) CUs out of the persistent backbone GEMM's budget.

    That kernel holds one workgroup on EVERY CU it may use for a whole launch (180-300 us at BASELINE configs[1]) and, with
    two backbone lanes, the next launch is already queued when one ends -- so without a reserve RCCL's kernels (gradient
    bucket all-reduce, SyncBN statistics, embedding all-gather: train.py:283-286 of the reference) start only when some
    GEMM workgroup happens to run out of tiles.  With the reserve they start at once; the GEMMs run on 248 of 256 CUs
    (-3 % of their throughput, measured in DESIGN.md section 5).  No-op without collectives.  Returns the GEMM's CU budget."""
    from .. import _lib
    if not collectives_active():
        return 0          # nothing to reserve for; a budget set through MVF_GEMM_CUS / mvf_gemm_tc_set_cus stays as it is
    reserve = int(os.environ.get('MVF_RCCL_CUS', '8'))
    dev = torch.cuda.current_device() if device is None else device
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    budget = max(8, cus - max(reserve, 0)) if reserve > 0 else 0
    _lib.call('mvf_gemm_tc_set_cus', budget)
    return budget


def get_world_size():
    return dist.get_world_size() if is_dist() else 1


def get_rank():
    return dist.get_rank() if is_dist() else 0


def is_root_proc():
    return get_rank() == 0


def synchronize():
    """Barrier across all ranks (distributed.py:136-148)."""
    if is_dist() and dist.get_world_size() > 1:
        dist.barrier()


def all_reduce(tensors, average=True):
    """In-place all-reduce of a list of tensors, optionally averaged (distributed.py:38-54)."""
    if not collectives_active():
        return tensors
    for t in tensors:
        dist.all_reduce(t, async_op=False)
    if average:
        ws = dist.get_world_size()
        for t in tensors:
            t.mul_(1.0 / ws)
    return tensors


def all_gather(tensors):
    """Non-differentiable tensor all-gather + cat(dim 0) (distributed.py:16-35)."""
    if not collectives_active():
        return tensors
    out = []
    for t in tensors:
        parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, t.contiguous())
        out.append(torch.cat(parts, dim=0))
    return out


class _GatherRows(torch.autograd.Function):
    """all_gather along dim 0 whose backward hands each rank the gradient rows of ITS slice.  Every rank
    evaluates the same global loss on the gathered rows, so d(global loss)/d(local rows) is just the local slice
    of the upstream gradient -- no reduce-scatter needed (the loss kernel only produces those rows anyway)."""

    @staticmethod
    def forward(ctx, x):
        ws, rk = dist.get_world_size(), dist.get_rank()
        parts = [torch.empty_like(x) for _ in range(ws)]
        dist.all_gather(parts, x.contiguous())
        ctx.rows, ctx.rank = x.shape[0], rk
        return torch.cat(parts, dim=0)

    @staticmethod
    def backward(ctx, g):
        return g[ctx.rank * ctx.rows:(ctx.rank + 1) * ctx.rows].contiguous()


def gather_rows(x):
    if not collectives_active():
        return x
    return _GatherRows.apply(x)


class FlatBuffers:
    """Re-homes a list of parameters into one flat fp32 buffer (and their .grad into a second one).

    `fuse_groups`: tuples of parameters that some op wants to see as ONE tensor (the Q|K|V weights / biases of a
    MultiheadedAttention): they are laid out back to back, in the given order, and `fused[group_index]` holds
    (parameter view, gradient view) over the whole group -- the op then needs no torch.cat in forward and its backward
    writes one gradient block.  Every parameter gets `p._mvf_grad` = its gradient slot (see ops.grad_slot)."""

    def __init__(self, params, fuse_groups=()):
        params = [p for p in params]
        assert params, 'no trainable parameters'
        ids = {id(p) for p in params}
        groups = [tuple(g) for g in fuse_groups if all(id(q) in ids for q in g)]
        member = {id(q): gi for gi, g in enumerate(groups) for q in g}
        # layout order: original order, but a fuse group is emitted whole at the position of its first member
        layout, seen = [], set()
        for p in params:
            if id(p) in seen:
                continue
            for q in (groups[member[id(p)]] if id(p) in member else (p,)):
                layout.append(q)
                seen.add(id(q))
        self.params = params                 # optimizer order (state_dict indices)
        dev = params[0].device
        # 4-element (16 B) alignment per tensor so every view can be read with float4
        off, n = {}, 0
        for p in layout:
            off[id(p)] = n
            n += (p.numel() + 3) // 4 * 4
        self.offsets = [off[id(p)] for p in params]
        self.numel = n
        self.flat_p = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(n, device=dev, dtype=torch.float32)
        for p, o in zip(self.params, self.offsets):
            v = self.flat_p[o:o + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
            p.grad = self.flat_g[o:o + p.numel()].view_as(p)
            p._mvf_grad = p.grad
            p._mvf_flat = self                      # ops.grad_slot marks the buffer dirty through this
            if p.requires_grad:
                p.register_hook(self._mark_dirty)   # gradients that arrive through autograd's AccumulateGrad
        # True while the gradient buffer may hold anything but zeros.  The fused Adam kernel zeroes the gradients it consumes,
        # so the zero_grad() that follows a step has nothing to do unless something wrote gradients in between.
        self.dirty = True
        self.fused = []
        for g in groups:
            o0, tot = off[id(g[0])], sum(q.numel() for q in g)
            contiguous = all(q.numel() % 4 == 0 for q in g)           # no alignment padding inside the group
            shape = (sum(q.shape[0] for q in g),) + tuple(g[0].shape[1:])
            self.fused.append((self.flat_p[o0:o0 + tot].view(shape), self.flat_g[o0:o0 + tot].view(shape)) if contiguous else None)

    def _mark_dirty(self, _grad):
        self.dirty = True

    def zero_grad(self, force=False):
        if self.dirty or force:
            self.flat_g.zero_()
            self.dirty = False
        for p, o in zip(self.params, self.offsets):   # re-attach in case something set .grad to None
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view_as(p)
                p._mvf_grad = p.grad


class GradReducer:
    """Bucketed asynchronous gradient all-reduce (SUM) over a FlatBuffers gradient buffer."""

    def __init__(self, flat, bucket_bytes=8 << 20, group=None):
        self.flat = flat
        self.group = group
        self.world = get_world_size()
        self.active = collectives_active()
        # buckets in REVERSE layout order (gradients become ready roughly output -> input); a bucket is a contiguous
        # element range of the flat gradient buffer
        self.buckets = []          # (start, end) element ranges
        self.bucket_of = {}
        cur_end = flat.numel
        cur_bytes = 0
        idxs = []
        order = sorted(range(len(flat.params)), key=lambda i: flat.offsets[i])
        for pos in reversed(range(len(order))):
            i = order[pos]
            idxs.append(i)
            cur_bytes += flat.params[i].numel() * 4
            if cur_bytes >= bucket_bytes or pos == 0:
                start = flat.offsets[i]
                b = len(self.buckets)
                self.buckets.append((start, cur_end))
                for j in idxs:
                    self.bucket_of[j] = b
                cur_end, cur_bytes, idxs = start, 0, []
        self.sizes = [sum(1 for j in self.bucket_of.values() if j == b) for b in range(len(self.buckets))]
        self.pending = None
        self.works = None
        # bench.py: `timing = True` records, per step, a device event pair around the wait for the bucket all-reduces --
        # the communication time the compute stream could NOT hide under backward (finish() is called right after the last
        # backward kernel has been enqueued and right before the optimizer's kernels)
        self.timing = False
        self.exposed = []          # (event before the wait, event after it)
        if self.active:
            for i, p in enumerate(flat.params):
                p.register_post_accumulate_grad_hook(self._make_hook(i, False))   # gradients that travel through autograd
                p._mvf_ready = self._make_hook(i, True)       # gradients a kernel wrote into the slot (ops.grad_ready)
        self.reset()

    def reset(self):
        self.pending = list(self.sizes)
        self.works = [None] * len(self.buckets)
        self.seen = set()

    def _launch(self, b):
        s, e = self.buckets[b]
        self.works[b] = dist.all_reduce(self.flat.flat_g[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _make_hook(self, i, from_kernel):
        def hook(_param):
            b = self.bucket_of[i]
            if i in self.seen:
                # One "final" signal per parameter and step.  The autograd engine runs a parameter's AccumulateGrad node --
                # and this hook -- even when the op's backward returned None for it (the kernel had written the slot and
                # signalled already): that late call accumulates nothing and is ignored.  A second KERNEL signal after the
                # bucket's all-reduce is in flight is a real bug: a contribution would be added to the slot during or after
                # the reduction (silently wrong, rank-dependent gradients).  No shipped module shares a parameter between two
                # kernel-accumulating ops; a future one must signal on its LAST use.
                if from_kernel and self.works[b] is not None:
                    raise RuntimeError('GradReducer: parameter #%d signalled again after its bucket\'s all-reduce was launched '
                                       '(a parameter used by two gradient-accumulating ops must signal on the last use)' % i)
                return
            self.seen.add(i)
            self.pending[b] -= 1
            if self.pending[b] == 0 and self.works[b] is None:
                self._launch(b)
        return hook

    def finish(self):
        """Launch whatever was not triggered (parameters without a gradient this step) and wait for all."""
        if self.active:
            for b in range(len(self.buckets)):
                if self.works[b] is None:
                    self._launch(b)
            ev = None
            if self.timing and self.flat.flat_g.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            for w in self.works:
                w.wait()
            if ev is not None:
                ev[1].record()
                self.exposed.append(ev)
        self.reset()
        return 1.0 / self.world   # scale that turns the SUM into DDP's average

    def bytes_per_step(self):
        """payload of the gradient all-reduce of one step (every bucket once): fp32 elements of the flat gradient buffer"""
        return 4 * sum(e - s for s, e in self.buckets)

    def exposed_ms(self):
        """mean device time per step between the end of backward and the start of the optimizer that the bucket all-reduces
        kept the compute stream waiting (call after a synchronize; clears the record)"""
        if not self.exposed:
            return None
        ms = sum(a.elapsed_time(b) for a, b in self.exposed) / len(self.exposed)
        self.exposed = []
        return ms
