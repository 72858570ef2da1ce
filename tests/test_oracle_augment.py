"""oracle/augment.py against the goldens of the imported reference (tests/golden/augment.npz, gen_golden.py `augment`):
crop-parameter draws, random_resized_crop, flip, grayscale, resize, normalisation, the validation pipeline and a
ComposeOp of the reference's own ops.  Colour jitter / blur (torchvision, absent here) have no reference golden: their
restatement is checked for the algebraic properties torchvision documents."""
import os
import random
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import _cases as C  # noqa: E402
from oracle import augment as A  # noqa: E402

G = np.load(os.path.join(HERE, 'golden', 'augment.npz'))
NOOP = dict(mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0))


def close(a, b, tol=1e-6):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert (a - b).abs().max().item() <= tol, (a - b).abs().max().item()


def test_crop_parameter_draws_follow_the_reference():
    for n, (hh, ww, seed) in enumerate(C.AUG_CROP_CASES):
        random.seed(seed)
        got = np.array([A.get_param_spatial_crop((0.8, 1.0), (3.0 / 4.0, 4.0 / 3.0), hh, ww) for _ in range(6)])
        assert (got == G['crop%d' % n]).all(), n
    fb = np.array([A.get_param_spatial_crop((0.8, 1.0), (3.0, 4.0), 40, 52), A.get_param_spatial_crop((0.8, 1.0), (0.1, 0.2), 40, 52)])
    assert (fb == G['crop_fallback']).all()


@pytest.mark.parametrize('n', range(len(C.AUG_CLIP_CASES)))
def test_reference_own_ops(n):
    t, hh, ww, size, seed = C.AUG_CLIP_CASES[n]
    x = C.aug_clip(t, hh, ww, seed)
    i, j, h, w = [int(v) for v in G['rrc%d_param' % n]]
    close(A.apply(x, A.Params(crop=(i, j, h, w), **NOOP), size), G['rrc%d' % n])
    close(A.apply(x, A.Params(**NOOP), size), G['resize%d' % n])
    close(A.ref_grayscale(x), G['gray%d' % n])
    close(torch.flip(x, dims=(-1,)), G['flip%d' % n])
    # normalisation alone: resize to the frame's own size is the identity only for square frames -> compare per op
    out = torch.empty_like(x)
    for c in range(3):
        out[:, c] = (x[:, c] - A.MEAN[c]) / A.STD[c]
    close(out, G['norm%d' % n])
    p = A.val_params(hh, ww, size)
    i, j, h, w = p.crop
    close(x[:, :, i:i + h, j:j + w], G['ucrop%d' % n])
    close(A.apply(x, p, size), G['val%d' % n])


@pytest.mark.parametrize('n', range(len(C.AUG_CLIP_CASES)))
def test_compose_of_reference_ops_draw_order(n):
    """ComposeOp([random_resized_crop, RandomOp(flip, .5), RandomOp(grayscale, .2), resize, normalise]): the oracle makes the
    same `random` draws in the same order (crop attempt loop, then one uniform per RandomOp)."""
    t, hh, ww, size, seed = C.AUG_CLIP_CASES[n]
    x = C.aug_clip(t, hh, ww, seed)
    random.seed(seed + 200)
    for k in range(4):
        p = A.Params()
        p.crop = A.get_param_spatial_crop((0.8, 1.0), (3.0 / 4.0, 4.0 / 3.0), hh, ww)
        p.flip = random.uniform(0, 1) < 0.5
        p.gray = random.uniform(0, 1) < 0.2
        close(A.apply(x, p, size), G['pipe%d' % n][k])


def test_ssl_draws_are_reproducible_and_in_range():
    random.seed(5)
    torch.manual_seed(5)
    ps = [A.draw_ssl_params(360, 480) for _ in range(200)]
    random.seed(5)
    torch.manual_seed(5)
    ps2 = [A.draw_ssl_params(360, 480) for _ in range(200)]
    assert all(a.crop == b.crop and a.color == b.color and a.blur == b.blur and a.gray == b.gray and a.flip == b.flip
               for a, b in zip(ps, ps2))
    assert 0.7 < np.mean([len(p.color) == 4 for p in ps]) < 0.9          # ColorJitterOp prob 0.8
    assert 0.3 < np.mean([p.blur is not None for p in ps]) < 0.5         # GaussianBlurOp prob 0.4
    assert 0.1 < np.mean([p.gray for p in ps]) < 0.3
    for p in ps:
        for op, f in p.color:
            lo, hi = ((-0.2, 0.2) if op == A.HUE else (0.2, 1.8))
            assert lo <= f <= hi
        assert sorted(op for op, _ in p.color) in ([], [0, 1, 2, 3])
        assert p.blur is None or 0.1 <= p.blur <= 2.0


def test_torchvision_restatement_properties():
    """Identities of the torchvision ops: factor 1 / hue 0 are the identity (up to rounding), factor 0 collapses to black /
    the frame's mean gray / the pixel's gray; a blur kernel sums to 1 and leaves a constant image unchanged."""
    x = C.aug_clip(2, 20, 28, 7)
    for op in (A.BRIGHTNESS, A.CONTRAST, A.SATURATION):
        close(A.color_step(x, op, 1.0), x, 1e-6)
    close(A.color_step(x, A.HUE, 0.0), x, 2e-6)
    close(A.color_step(x, A.HUE, 1.0), x, 2e-6)                      # hue is periodic
    close(A.color_step(x, A.BRIGHTNESS, 0.0), torch.zeros_like(x))
    g = A.tv_gray(x)
    close(A.color_step(x, A.SATURATION, 0.0), g.expand_as(x), 1e-6)
    close(A.color_step(x, A.CONTRAST, 0.0), g.mean(dim=(-3, -2, -1), keepdim=True).expand_as(x), 1e-6)
    hsv = A.rgb2hsv(x)
    assert hsv[:, 0].min() >= 0 and hsv[:, 0].max() < 1.0 + 1e-6
    close(A.hsv2rgb(hsv), x, 2e-6)
    k = A.gaussian_kernel1d(9, 1.3)
    assert abs(float(k.sum()) - 1.0) < 1e-6 and torch.equal(k, k.flip(0))
    c = torch.full((1, 3, 12, 14), 0.37)
    close(A.gaussian_blur(c, (5, 9), 0.8), c, 1e-6)
    imp = torch.zeros(1, 3, 21, 21)
    imp[:, :, 10, 10] = 1.0
    b = A.gaussian_blur(imp, (5, 9), 1.1)
    close(b[0, 0, 6:15, 8:13], torch.outer(A.gaussian_kernel1d(9, 1.1), A.gaussian_kernel1d(5, 1.1)), 1e-7)   # 9 rows x 5 columns
