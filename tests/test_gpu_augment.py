"""View augmentation: host-side draws vs the oracle (CPU), HIP kernels vs the oracle through the C ABI
(GPU).  Tolerance: 2e-5 absolute on the NORMALISED output (the last step divides by std ~ 0.225, so ~4e-6 before it);
the colour/blur steps follow torchvision's formulas op by op (oracle/augment.py: unpinned by the reference for those two)."""
import os
import random
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import _cases as C  # noqa: E402
from oracle import augment as A  # noqa: E402
from video_rep_learning_amd.datasets import augment as P  # noqa: E402
from video_rep_learning_amd.utils import presets  # noqa: E402

TOL = 2e-5


def to_oracle(p):
    """MvfAugmentParams -> oracle Params."""
    return A.Params(crop=(p.crop_top, p.crop_left, p.crop_h, p.crop_w), flip=p.flip,
                    color=[(p.color_op[k], p.color_factor[k]) for k in range(p.n_color)],
                    blur=p.blur_sigma if p.blur_sigma > 0 else None, gray=p.gray, ksize=(p.blur_kx, p.blur_ky),
                    mean=tuple(p.mean), std=tuple(p.std))


def test_host_draws_equal_oracle_draws():
    cfg = presets.baseline_config_2()
    pol = P.SSLAugment(cfg)
    for seed, (h, w) in enumerate([(360, 480), (480, 270), (224, 224), (40, 52)]):
        random.seed(seed)
        torch.manual_seed(seed)
        got = [to_oracle(pol.draw(h, w)) for _ in range(50)]
        random.seed(seed)
        torch.manual_seed(seed)
        ref = [A.draw_ssl_params(h, w, strength=cfg.AUGMENTATION.STRENGTH) for _ in range(50)]
        for a, b in zip(got, ref):
            assert a.crop == tuple(b.crop) and a.flip == b.flip and a.gray == b.gray
            assert [o for o, _ in a.color] == [o for o, _ in b.color]
            assert all(abs(x - y) < 1e-7 for (_, x), (_, y) in zip(a.color, b.color))
            assert (a.blur is None) == (b.blur is None) and (a.blur is None or abs(a.blur - b.blur) < 1e-7)
    v = P.ValPreprocess(cfg)
    for h, w in [(360, 480), (480, 270), (224, 224), (300, 224)]:
        assert to_oracle(v.draw(h, w)).crop == A.val_params(h, w, cfg.IMAGE_SIZE).crop
    with pytest.raises(NotImplementedError):
        P.create_data_augment(cfg, augment=True)


gpu = pytest.mark.gpu


def run(x, plist, size):
    from video_rep_learning_amd import ops
    return ops.augment_clips(x.cuda(), plist, size).cpu()


def check(x, plist, size, tol=TOL):
    got = run(x, plist, size)
    for k, p in enumerate(plist):
        ref = A.apply(x[k], to_oracle(p), size)
        err = (got[k] - ref).abs().max().item()
        assert err <= tol, (k, err, to_oracle(p).__dict__)
    return got


@gpu
@pytest.mark.parametrize('case', range(len(C.AUG_CLIP_CASES)))
def test_single_steps_and_orders(case):
    t, h, w, size, seed = C.AUG_CLIP_CASES[case]
    x = C.aug_clip(t, h, w, seed)
    whole = (0, 0, h, w)
    plist = [P._params(whole), P._params((3, 5, h - 7, w - 9)), P._params(whole, flip=True), P._params(whole, gray=True),
             P._params(whole, sigma=0.1), P._params(whole, sigma=0.77), P._params(whole, sigma=2.0)]
    for op, fs in ((0, (0.2, 1.0, 1.8)), (1, (0.2, 1.0, 1.8)), (2, (0.2, 1.0, 1.8)), (3, (-0.2, 0.0, 0.13, 0.2))):
        plist += [P._params(whole, color=[(op, f)]) for f in fs]
    import itertools
    fac = {0: 1.3, 1: 0.6, 2: 1.7, 3: -0.11}
    for order in itertools.permutations(range(4)):                     # contrast first, in the middle, last
        plist.append(P._params((1, 2, h - 3, w - 4), flip=order[0] % 2, color=[(o, fac[o]) for o in order],
                               sigma=0.9 if order[1] == 2 else 0.0, gray=order[2] == 0))
    for k in range(0, len(plist), 8):                                  # batches of <= 8 and a ragged last one
        check(x.unsqueeze(0).expand(len(plist[k:k + 8]), *x.shape).contiguous(), plist[k:k + 8], size)


@gpu
def test_random_draws_against_oracle_and_determinism():
    cfg = presets.make_cfg(image_size=32)
    pre = P.get_data_preprocess(cfg, 'train')
    assert isinstance(pre.policy, P.SSLAugment)
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 256, (11, 3, 3, 45, 61), generator=g).float() / 255.0   # 11 clips: two launches (8 + 3)
    random.seed(7)
    torch.manual_seed(7)
    plist = [pre.policy.draw(45, 61) for _ in range(11)]
    got = check(x, plist, 32)
    assert torch.equal(got, run(x, plist, 32))                         # fixed-order reductions: bitwise repeatable
    # the callable forms: per clip (reference signature) and preproc_views' interleaved order
    random.seed(7)
    torch.manual_seed(7)
    one = pre(x[0].cuda()).cpu()
    assert torch.equal(one, got[0])
    random.seed(7)
    torch.manual_seed(7)
    both = P.preproc_views(x[0:4].cuda(), x[4:8].cuda(), pre).cpu()    # draws: v0[0], v1[0], v0[1], v1[1], ...
    random.seed(7)
    torch.manual_seed(7)
    pl2 = [pre.policy.draw(45, 61) for _ in range(8)]
    inter = torch.stack([x[0:4], x[4:8]], 1).reshape(8, 3, 3, 45, 61)
    assert torch.equal(both.reshape(8, 3, 3, 32, 32), run(inter, pl2, 32))


@gpu
def test_full_size_clip_and_validation_path():
    """BASELINE-sized input: a 32-frame 360x480 clip -> 224; SSL draws and the validation (centre-crop) path."""
    cfg = presets.baseline_config_2()
    g = torch.Generator().manual_seed(5)
    x = torch.randint(0, 256, (2, 32, 3, 360, 480), generator=g).float() / 255.0
    random.seed(11)
    torch.manual_seed(11)
    pol = P.SSLAugment(cfg)
    plist = [pol.draw(360, 480), pol.draw(360, 480)]
    plist[0] = P._params((plist[0].crop_top, plist[0].crop_left, plist[0].crop_h, plist[0].crop_w), flip=True,
                         color=[(3, 0.07), (1, 1.4), (0, 0.8), (2, 1.2)], sigma=1.3, gray=False)   # every step at once
    check(x, plist, 224)
    val = P.get_data_preprocess(cfg, 'val')
    vp = [val.policy.draw(360, 480)] * 2
    got = check(x, vp, 224)
    # centre crop without scaling: pure normalisation of the window
    y0, x0 = (360 - 224 + 1) // 2, (480 - 224 + 1) // 2
    ref = (x[:, :, :, y0:y0 + 224, x0:x0 + 224] - torch.tensor(A.MEAN).view(1, 1, 3, 1, 1)) / torch.tensor(A.STD).view(1, 1, 3, 1, 1)
    assert (got - ref).abs().max().item() <= 1e-6


@gpu
def test_bad_parameters_are_refused():
    from video_rep_learning_amd import ops, _lib
    x = torch.rand(1, 2, 3, 20, 20).cuda()
    with pytest.raises(_lib.MvfError):
        ops.augment_clips(x, [P._params((0, 0, 21, 20))], 8)                           # window outside the frame
    with pytest.raises(_lib.MvfError):
        ops.augment_clips(x, [P._params((0, 0, 20, 20), color=[(1, 1.0), (1, 0.5)])], 8)   # a step twice
    with pytest.raises(_lib.MvfError):
        ops.augment_clips(x, [P._params((0, 0, 20, 20), sigma=1.0)], 4)                # reflect pad 4 >= size 4
    with pytest.raises(_lib.MvfError):
        ops.augment_clips(x.cpu(), [P._params((0, 0, 20, 20))], 8)                     # no CPU fallback
