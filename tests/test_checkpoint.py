"""Checkpoint-format compatibility (reference CARL_MVF/models/__init__.py:17-60).

A reference checkpoint is {'epoch', 'model_state' (keys of TransformerModel.state_dict()), 'optimizer_state'
(torch.optim.Adam.state_dict() over the two param groups of utils/optimizer.py:26-66)}.  These tests run on CPU (no
kernel is launched): key names / shapes against the golden state dict that tests/golden/gen_golden.py took from the
imported reference modules, FusedAdam <-> torch.optim.Adam state interchange, and a save/load round trip."""
import os

import pytest
import torch

from video_rep_learning_amd.models import build_model, save_checkpoint, load_checkpoint
from video_rep_learning_amd.utils import presets
from video_rep_learning_amd.utils.optimizer import construct_optimizer, select_parameters

HERE = os.path.dirname(os.path.abspath(__file__))


def small_cfg():
    cfg = presets.baseline_config_2(compute_dtype='fp32')
    return cfg


@pytest.fixture(scope='module')
def model_cfg():
    cfg = small_cfg()
    torch.manual_seed(3)
    return build_model(cfg, 0), cfg


def test_state_dict_keys_match_reference_golden(model_cfg):
    """state_keys.json: keys/shapes of the imported reference modules at BASELINE config #2 and the parameter order of
    the reference's two optimizer groups (gen_golden.py gen_state_keys)."""
    import json
    model, cfg = model_cfg
    gold = json.load(open(os.path.join(HERE, 'golden', 'state_keys.json')))
    ref = {k: tuple(v) for k, v in gold['state_dict'].items()}
    assert len(ref) > 40
    ours = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.startswith('backbone.')}
    assert set(ours) == set(ref), (sorted(set(ours) ^ set(ref)))
    for k in ref:
        assert ref[k] == ours[k], (k, ref[k], ours[k])
    # optimizer_state['state'] is numbered by position in (bn group, non-bn group): the order must be the reference's
    names = {id(p): n for n, p in model.named_parameters()}
    bn, non_bn = select_parameters(model, cfg)
    assert [names[id(p)] for p in bn] == gold['optimizer_groups'][0]
    assert [names[id(p)] for p in non_bn] == gold['optimizer_groups'][1]
    # the frozen timm ViT sits behind FeatureExtractor.model (transformer.py:306-312): backbone.model.<timm name>
    bk = [k for k in model.state_dict() if k.startswith('backbone.')]
    assert all(k.startswith('backbone.model.') for k in bk)
    for k in ('backbone.model.cls_token', 'backbone.model.pos_embed', 'backbone.model.patch_embed.proj.weight',
              'backbone.model.blocks.11.attn.qkv.weight', 'backbone.model.blocks.0.mlp.fc2.bias', 'backbone.model.norm.weight'):
        assert k in model.state_dict(), k


@pytest.mark.parametrize('variant', ['split', 'cls_res'])
def test_state_dict_keys_of_split_backbone_and_cls_res_match_reference(variant):
    """glue_split.npz '<variant>_keys': state-dict keys of the imported reference TransformerModel with LAYER = 10
    (backbone.model.*, backbone.blocks.<i> aliases, res_finetune.model.blocks.<i - 10>.*, res_finetune.model.norm.*) and
    with MODEL.CLS_RES (cls_res_res.*): checkpoints of either kind load key for key."""
    import numpy as np
    from video_rep_learning_amd.utils import presets
    gold = set(np.load(os.path.join(HERE, 'golden', 'glue_split.npz'))[variant + '_keys'].tolist())
    cfg = presets.make_cfg(network='TIMM-vit_small_patch16_224.dino', num_frames=8, batch_size=2, image_size=32,
                           SMART_FEATS='10,11' if variant == 'split' else '3,7,11', NUM_LAYERS=2)   # the golden's head has 2 encoder layers
    if variant == 'split':
        cfg.MODEL.BASE_MODEL.LAYER = 10
    else:
        cfg.MODEL.CLS_RES = True
    ours = set(build_model(cfg, 0).state_dict().keys())
    assert ours == gold, sorted(ours ^ gold)[:20]


def _torch_adam(model, cfg):
    bn, non_bn = select_parameters(model, cfg)
    wd = cfg.OPTIMIZER.WEIGHT_DECAY
    return torch.optim.Adam([{'params': bn, 'weight_decay': wd}, {'params': non_bn, 'weight_decay': wd}],
                            lr=cfg.OPTIMIZER.LR.INITIAL_LR, betas=(0.9, 0.999), weight_decay=wd)


def test_optimizer_state_interchanges_with_torch_adam(model_cfg):
    model, cfg = model_cfg
    ref = _torch_adam(model, cfg)
    g = torch.Generator().manual_seed(9)
    for grp in ref.param_groups:
        for p in grp['params']:
            p.grad = torch.randn(p.shape, generator=g) * 1e-3
    before = {id(p): p.detach().clone() for grp in ref.param_groups for p in grp['params']}
    ref.step()
    ref.step()
    for grp in ref.param_groups:                       # the model must not keep the stand-in's updates
        for p in grp['params']:
            p.data.copy_(before[id(p)])
            p.grad = None
    rsd = ref.state_dict()

    opt = construct_optimizer(model, cfg)              # FusedAdam over flat buffers (CPU here: no step is taken)
    osd0 = opt.state_dict()
    assert [g_['params'] for g_ in osd0['param_groups']] == [g_['params'] for g_ in rsd['param_groups']]
    for a, b in zip(osd0['param_groups'], rsd['param_groups']):
        for k in ('lr', 'betas', 'eps', 'weight_decay'):
            assert a[k] == b[k], k
    opt.load_state_dict(rsd)                            # a reference checkpoint's optimizer state
    assert opt.step_count == 2
    osd = opt.state_dict()
    assert set(osd['state']) == set(rsd['state'])
    for i, st in rsd['state'].items():
        assert float(osd['state'][i]['step']) == float(st['step'])
        assert torch.equal(osd['state'][i]['exp_avg'], st['exp_avg']), i
        assert torch.equal(osd['state'][i]['exp_avg_sq'], st['exp_avg_sq']), i
    ref2 = _torch_adam(model, cfg)                      # and back: torch.optim.Adam accepts FusedAdam's state dict
    ref2.load_state_dict(osd)
    for i, st in rsd['state'].items():
        p = ref2.param_groups[0]['params'][i] if i < len(ref2.param_groups[0]['params']) else \
            ref2.param_groups[1]['params'][i - len(ref2.param_groups[0]['params'])]
        assert torch.equal(ref2.state[p]['exp_avg'], st['exp_avg'])


def test_checkpoint_round_trip(model_cfg, tmp_path):
    model, cfg = model_cfg
    cfg = type(cfg)(cfg)
    cfg.LOGDIR = str(tmp_path)
    opt = construct_optimizer(model, cfg)
    opt.exp_avg.normal_(generator=torch.Generator().manual_seed(1))
    opt.exp_avg_sq.uniform_(generator=torch.Generator().manual_seed(2))
    opt.step_count = 7
    save_checkpoint(cfg, model, opt, 4)
    ck = torch.load(os.path.join(str(tmp_path), 'checkpoints', 'checkpoint_epoch_00004.pth'), map_location='cpu',
                    weights_only=False)
    assert {'epoch', 'model_state', 'optimizer_state'} <= set(ck) and ck['epoch'] == 4
    assert set(ck['optimizer_state']) == {'state', 'param_groups'}

    torch.manual_seed(11)
    model2 = build_model(cfg, 0)
    opt2 = construct_optimizer(model2, cfg)
    assert load_checkpoint(cfg, model2, opt2) == 5       # resumes at the next epoch (models/__init__.py:46)
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k
    assert opt2.step_count == 7
    # moments land at the same PARAMETER (flat-buffer layouts may differ between the two optimizers' fuse groups)
    sa, sb = opt.state_dict()['state'], opt2.state_dict()['state']
    for i in sa:
        assert torch.equal(sa[i]['exp_avg'], sb[i]['exp_avg']) and torch.equal(sa[i]['exp_avg_sq'], sb[i]['exp_avg_sq']), i
    # parameters still alias the flat buffer after load_state_dict (the optimizer keeps stepping the live tensors)
    p0 = opt2.flat.params[0]
    assert p0.data_ptr() >= opt2.flat.flat_p.data_ptr() and p0.data_ptr() < opt2.flat.flat_p.data_ptr() + opt2.flat.flat_p.numel() * 4
