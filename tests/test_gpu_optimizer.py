"""FusedAdam bookkeeping around skipped (non-finite) steps: state_dict() is side-effect free, load_state_dict() leaves no stale
skipped-step count behind, and the finite-gradient guard does not depend on clipping being on."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from video_rep_learning_amd.utils.optimizer import FusedAdam  # noqa: E402

DEV = 'cuda'


def _make(seed=0):
    g = torch.Generator().manual_seed(seed)
    ps = [torch.nn.Parameter(torch.randn(33, 17, generator=g).to(DEV)), torch.nn.Parameter(torch.randn(64, generator=g).to(DEV))]
    return ps, FusedAdam([{'params': ps}], lr=1e-2, weight_decay=1e-5)


def _grads(opt, ps, seed, bad=False):
    g = torch.Generator().manual_seed(seed)
    opt.zero_grad()
    for p in ps:
        p.grad.copy_(torch.randn(p.shape, generator=g).to(DEV))
    if bad:
        ps[0].grad.view(-1)[5] = float('nan')


@pytest.mark.parametrize('clip', [10.0, 0.0])
def test_skipped_steps_state_dict_and_reload(clip):
    ps, opt = _make()
    ref_ps, ref = _make()                     # the same trajectory without the bad step
    for k, bad in enumerate([False, True, False]):
        _grads(opt, ps, 100 + k, bad)
        before = [p.detach().clone() for p in ps]
        opt.step(max_norm=clip)
        if bad:                                # guard active with and without clipping: nothing moved
            assert all(torch.equal(a, p.detach()) for a, p in zip(before, ps))
        else:
            _grads(ref, ref_ps, 100 + k)
            ref.step(max_norm=clip)
    assert opt.skipped_steps() == 1 and opt.step_count == 3
    for a, b in zip(ps, ref_ps):               # bias correction used the EFFECTIVE step count
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
    sd = opt.state_dict()
    assert float(sd['state'][0]['step']) == 2.0
    assert opt.skipped_steps() == 1 and opt.step_count == 3      # no side effects
    sd2 = opt.state_dict()
    assert float(sd2['state'][0]['step']) == 2.0
    # reload into the SAME optimizer: the stale device counter must not be taken off again
    opt.load_state_dict(sd)
    assert opt.step_count == 2 and opt.skipped_steps() == 0
    _grads(opt, ps, 200)
    _grads(ref, ref_ps, 200)
    opt.step(max_norm=clip)
    ref.step(max_norm=clip)
    for a, b in zip(ps, ref_ps):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
