"""Synthetic clip source with the reference's batch contract.

The reference's loaders (CARL_MVF/datasets/penn_action.py:101-130, collated by the default collate) yield
    videos=(view0, view1) each [B,T,3,H,W] float, labels [B,2,T], seq_lens [B,2] int64,
    chosen_steps [B,2,T] int64 (sorted, clamped to [0, L-1]), video_masks [B,2,T] float {0,1}, names
Video decode + CPU sampling are out of scope; this generator reproduces the tuple with
N(0,1) frames standing for mean/std-normalised images: seed 1234 + rank, seq_len 100, sorted random
steps, and an optional padded video (seq_len < T: trailing frames masked, steps clamped like penn_action.py:180-197).
"""
import torch


class SyntheticClips:
    def __init__(self, batch_size, num_frames, image_size=224, iters=50, seed=1234, device='cpu', seq_len=100,
                 pad_every=0, resident=True, raw_hw=None):
        """raw_hw = (H, W): emit frames as the datasets do BEFORE pre-processing (penn_action.py:110-111: uint8 / 255, at
        the video's own size) so that the GPU-side augmentation (datasets/augment.py) is part of the loop."""
        self.b, self.t, self.s = batch_size, num_frames, image_size
        self.raw_hw = raw_hw
        self.iters, self.seed, self.device = iters, seed, device
        self.seq_len, self.pad_every, self.resident = seq_len, pad_every, resident
        self.sampler = None
        self.batch_sampler = None
        self._cache = None

    def __len__(self):
        return self.iters

    def _make(self, it):
        g = torch.Generator().manual_seed(self.seed + 7919 * it)
        b, t, s = self.b, self.t, self.s
        if self.raw_hw is None:
            views = [torch.randn(b, t, 3, s, s, generator=g) for _ in range(2)]
        else:
            views = [torch.randint(0, 256, (b, t, 3) + tuple(self.raw_hw), generator=g, dtype=torch.uint8).float() / 255.0
                     for _ in range(2)]
        seq_lens = torch.full((b, 2), self.seq_len, dtype=torch.long)
        steps = torch.sort(torch.randint(0, self.seq_len, (b, 2, t), generator=g), dim=-1)[0]
        masks = torch.ones(b, 2, t)
        if self.pad_every and it % self.pad_every == 0:
            L = max(2, (t * 5) // 8)
            seq_lens[0] = L
            steps[0] = torch.arange(t).clamp(max=L - 1)
            masks[0, :, L:] = 0
        labels = torch.zeros(b, 2, t, dtype=torch.long)
        names = ['synthetic_%d_%d' % (it, i) for i in range(b)]
        dev = self.device
        return ((views[0].to(dev), views[1].to(dev)), labels, seq_lens, steps, masks, names)

    def __iter__(self):
        for it in range(self.iters):
            if self.resident:        # one resident batch re-used (bench: inputs already in HBM)
                if self._cache is None:
                    self._cache = self._make(0)
                yield self._cache
            else:
                yield self._make(it)


def construct_dataloader(cfg, split, device='cpu', iters=None, rank=0, resident=False, raw_hw=None):
    loader = SyntheticClips(cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.NUM_FRAMES, cfg.IMAGE_SIZE,
                            iters=iters if iters is not None else 50, seed=1234 + rank + (0 if split == 'train' else 10 ** 6),
                            device=device, pad_every=4 if split == 'train' else 0, resident=resident, raw_hw=raw_hw)
    return loader, None


def get_data_preprocess(cfg, split, raw=False):
    """Normalised synthetic frames need no pre-processing (None); raw ones go through the reference's pipeline for the split
    (datasets/augment.py: get_data_preprocess)."""
    if not raw:
        return None
    from . import augment
    return augment.get_data_preprocess(cfg, split)


def preproc_views(view_0, view_1, data_preprocess, device):
    """train.preproc_views (train.py:39-53): -> [B, 2, T, 3, S, S] on `device`."""
    view_0, view_1 = view_0.to(device, non_blocking=True), view_1.to(device, non_blocking=True)
    if data_preprocess is None:
        return torch.stack([view_0, view_1], dim=1)
    from . import augment
    return augment.preproc_views(view_0, view_1, data_preprocess)
