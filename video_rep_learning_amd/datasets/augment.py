"""GPU-side view pre-processing with the reference's surface (CARL_MVF/datasets/data_augment.py:372-469,
`get_data_preprocess(cfg, mode)` -> callable applied per clip by train.preproc_views, train.py:39-53).

The random decisions are drawn on the host, by the same calls, in the same order and from the same generators (Python's
`random`, torch's global CPU generator) as the reference's op objects make them, so a seeded run augments the same way;
the pixel work of a whole batch of clips is three HIP launches (csrc/augment.hip) instead of a Python loop over
clips x ops.  Out of scope: `create_data_augment(cfg, augment=True)` (the non-SSL training pipeline, :416-441), whose
jitters round-trip every frame through 8-bit PIL images on the CPU; it raises NotImplementedError."""
import math
import random

import torch

from .. import ops
from .._lib import MvfAugmentParams

BRIGHTNESS, CONTRAST, SATURATION, HUE = 0, 1, 2, 3
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def _get_param_spatial_crop(scale, ratio, height, width):
    """data_augment.py:254-284: up to 10 (area, log-ratio) draws, then the centre-crop fallback."""
    for _ in range(10):
        area = height * width
        target_area = random.uniform(*scale) * area
        log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
        aspect_ratio = math.exp(random.uniform(*log_ratio))
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if 0 < w <= width and 0 < h <= height:
            i = random.randint(0, height - h)
            j = random.randint(0, width - w)
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def _params(crop, flip=False, color=(), sigma=0.0, gray=False):
    p = MvfAugmentParams()
    p.crop_top, p.crop_left, p.crop_h, p.crop_w = crop
    p.flip = int(bool(flip))
    p.n_color = len(color)
    for k, (op, f) in enumerate(color):
        p.color_op[k], p.color_factor[k] = int(op), float(f)
    p.blur_kx, p.blur_ky, p.blur_sigma = 5, 9, float(sigma)      # GaussianBlur(kernel_size=(5, 9)), data_augment.py:360
    p.gray = int(bool(gray))
    for c in range(3):
        p.mean[c], p.std[c] = MEAN[c], STD[c]
    return p


class SSLAugment:
    """create_ssl_data_augment(cfg, augment=True), data_augment.py:372-413."""

    def __init__(self, cfg):
        self.size = cfg.IMAGE_SIZE
        self.strength = cfg.AUGMENTATION.STRENGTH

    def draw(self, height, width):
        crop = _get_param_spatial_crop((0.8, 1.0), (3.0 / 4.0, 4.0 / 3.0), height, width)   # AugmentOp(random_resized_crop)
        flip = random.uniform(0, 1) < 0.5                                                   # RandomOp(flip, 0.5)
        color = []
        if random.uniform(0, 1) < 0.8:                                                      # ColorJitterOp(0.8, ...)
            s = self.strength
            rng = {BRIGHTNESS: (max(0.0, 1 - 0.8 * s), 1 + 0.8 * s), CONTRAST: (max(0.0, 1 - 0.8 * s), 1 + 0.8 * s),
                   SATURATION: (max(0.0, 1 - 0.8 * s), 1 + 0.8 * s), HUE: (-0.2 * s, 0.2 * s)}
            order = torch.randperm(4)                                                       # ColorJitter.get_params
            fac = {op: (float(torch.empty(1).uniform_(rng[op][0], rng[op][1])) if s != 0 else None)
                   for op in (BRIGHTNESS, CONTRAST, SATURATION, HUE)}
            color = [(int(op), fac[int(op)]) for op in order if fac[int(op)] is not None]
        sigma = 0.0
        if random.uniform(0, 1) < 0.4:                                                      # GaussianBlurOp(0.4)
            sigma = torch.empty(1).uniform_(0.1, 2.0).item()                                # GaussianBlur.get_params
        gray = random.uniform(0, 1) < 0.2                                                   # RandomOp(grayscale, 0.2)
        return _params(crop, flip, color, sigma, gray)


class ValPreprocess:
    """create_data_augment(cfg, augment=False), data_augment.py:442-456: centre uniform_crop (when
    AUGMENTATION.RANDOM_CROP), resize, normalise."""

    def __init__(self, cfg):
        self.size = cfg.IMAGE_SIZE
        self.crop = bool(cfg.AUGMENTATION.RANDOM_CROP)

    def draw(self, height, width):
        if not self.crop:
            return _params((0, 0, height, width))
        y = max(int(math.ceil((height - self.size) / 2)), 0)
        x = max(int(math.ceil((width - self.size) / 2)), 0)
        return _params((y, x, min(self.size, height - y), min(self.size, width - x)))


class ClipPreprocess:
    """Callable with the reference's per-clip signature ([T,3,H,W] -> [T,3,S,S]); `batch` processes many clips at once."""

    def __init__(self, policy):
        self.policy = policy

    def batch(self, clips):
        """clips [n, T, 3, H, W] on the device; one parameter draw per clip, in clip order."""
        n, _t, _c, h, w = clips.shape
        return ops.augment_clips(clips, [self.policy.draw(h, w) for _ in range(n)], self.policy.size)

    def __call__(self, clip):
        return self.batch(clip.unsqueeze(0))[0]


def create_ssl_data_augment(cfg, augment):
    return ClipPreprocess(SSLAugment(cfg) if augment else ValPreprocess(cfg))


def create_data_augment(cfg, augment):
    if augment:
        raise NotImplementedError('create_data_augment(augment=True): the non-SSL training jitters run through 8-bit PIL '
                                  'images on the CPU (data_augment.py:120-215); not on the MI355X path')
    return ClipPreprocess(ValPreprocess(cfg))


def get_data_preprocess(cfg, mode):
    """data_augment.py:462-469."""
    if cfg.SSL and mode == 'train':
        return create_ssl_data_augment(cfg, augment=True)
    if mode == 'train':
        return create_data_augment(cfg, augment=True)
    return create_data_augment(cfg, augment=False)


def preproc_views(view_0, view_1, data_preprocess):
    """train.preproc_views (train.py:39-53): views [B, T, 3, H, W] -> [B, 2, T, 3, S, S].  The reference draws view 0's and
    view 1's parameters alternately, video by video; the same order is kept here, then everything is one batch."""
    b = view_0.shape[0]
    inter = torch.stack([view_0, view_1], dim=1).reshape((2 * b,) + tuple(view_0.shape[1:]))   # v0[0], v1[0], v0[1], ...
    out = data_preprocess.batch(inter)
    return out.view((b, 2) + tuple(out.shape[1:]))
