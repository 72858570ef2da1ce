"""Host logic: config defaults / YAML merge / --opts typing behave like CARL_MVF/utils/{config,parser}.py."""
import glob
import os

import pytest
import yaml

from video_rep_learning_amd.utils import presets
from video_rep_learning_amd.utils.config import get_cfg, EasyDict
from video_rep_learning_amd.utils.parser import parse_args, load_config, convert_value, to_dict

REF = '/root/reference/CARL_MVF'
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason='reference not mounted (GPU box)')


def test_easydict_semantics():
    d = EasyDict({'A': {'B': 1}, 'L': [{'x': 1}]})
    assert d.A.B == 1 and d['A']['B'] == 1 and d.L[0].x == 1
    d.update({'A': {'C': 2}})          # shallow: replaces the sub-tree
    assert 'B' not in d.A and d.A.C == 2
    d.Z = {'q': 3}
    assert isinstance(d.Z, EasyDict) and d.Z.q == 3
    with pytest.raises(AttributeError):
        d.missing


def test_convert_value_typing():
    assert convert_value(True, 'false') is False and convert_value(False, 'True') is True
    assert convert_value(3, '7') == 7 and convert_value(0.1, '2') == 2.0 and convert_value('a', '3,7,11') == '3,7,11'
    assert convert_value([1, 2], '[3 4 5]') == [3, 4, 5]


def test_opts_override_and_eval_sync():
    args = parse_args(['--opts', 'TRAIN.NUM_FRAMES', '32', 'TRAIN.BATCH_SIZE', '4', 'SCL.NEGATIVE_TYPE', 'batch_noself'])
    cfg = load_config(args)
    assert cfg.TRAIN.NUM_FRAMES == 32 and cfg.EVAL.NUM_FRAMES == 32 and cfg.EVAL.BATCH_SIZE == 4
    assert cfg.SCL.NEGATIVE_TYPE == 'batch_noself'
    assert cfg.LOGDIR.startswith('/tmp/')
    with pytest.raises(KeyError):
        load_config(parse_args(['--opts', 'TRAIN.NOPE', '1']))


def test_local_rank_spellings(monkeypatch):
    assert parse_args(['--local_rank', '3']).local_rank == 3
    assert parse_args(['--local-rank', '2']).local_rank == 2
    monkeypatch.setenv('LOCAL_RANK', '5')
    assert parse_args([]).local_rank == 5


@needs_ref
def test_defaults_match_reference_config_py():
    """The default tree equals the reference's (parsed from its config.py without importing easydict)."""
    import types, sys, importlib.util
    ed = types.ModuleType('easydict')
    ed.EasyDict = EasyDict
    sys.modules['easydict'] = ed
    try:
        spec = importlib.util.spec_from_file_location('ref_config', os.path.join(REF, 'utils', 'config.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        del sys.modules['easydict']
    assert to_dict(mod.CONFIG) == to_dict(get_cfg())


@needs_ref
def test_preset_equals_penn_mvf_yaml():
    with open(os.path.join(REF, 'configs_mvf', 'penn_mvf.yml')) as f:
        ref = yaml.safe_load(f)
    assert ref == presets.penn_mvf()


@needs_ref
@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(REF, 'configs_mvf', '*.yml'))))
def test_reference_configs_drop_in(path):
    cfg = load_config(parse_args(['--cfg_file', path, '--logdir', '/tmp/x']))
    assert cfg.TRAINING_ALGO == 'scl' and 'EMBEDDER_MODEL' in cfg.MODEL
    assert cfg.EVAL.NUM_FRAMES == cfg.TRAIN.NUM_FRAMES
    assert 'TCC' in cfg and 'OPTIMIZER' in cfg          # untouched default sub-trees survive


def test_backbone_weights_with_another_image_size_are_resampled(tmp_path):
    """MODEL.BASE_MODEL.WEIGHTS stored for a different image size (DINOv2: 518 px): the position embedding's grid part is
    resampled (bicubic, antialias) like timm's resample_abs_pos_embed, the CLS entry kept; same size = untouched."""
    import torch
    from video_rep_learning_amd.models import vit
    big = vit.create_model('vit_small_patch14_dinov2.lvd142m', img_size=70, seed=3)            # 5 x 5 patches
    path = str(tmp_path / 'w.pth')
    torch.save({'model': big.state_dict()}, path)
    small = vit.create_model('vit_small_patch14_dinov2.lvd142m', weights=path, img_size=42)    # 3 x 3 patches
    assert small.pos_embed.shape == (1, 10, 384)
    assert torch.equal(small.pos_embed[:, 0], big.pos_embed[:, 0])
    assert torch.equal(small.blocks[3].mlp.fc1.weight, big.blocks[3].mlp.fc1.weight)
    same = vit.create_model('vit_small_patch14_dinov2.lvd142m', weights=path, img_size=70)
    assert torch.equal(same.pos_embed, big.pos_embed)
    # a constant grid stays constant under resampling; a linear ramp keeps its end-to-end ordering
    pe = torch.cat([torch.zeros(1, 1, 4), torch.full((1, 25, 4), 0.37)], 1)
    out = vit.resample_abs_pos_embed(pe, 10)
    assert torch.allclose(out[:, 1:], torch.full((1, 9, 4), 0.37), atol=1e-6)


def test_dinov2_giant_is_refused_loudly():
    """timm's DINOv2-giant has a SwiGLU MLP; the GELU kernels would compute another network."""
    import pytest
    from video_rep_learning_amd.models import build_model
    from video_rep_learning_amd.utils import presets
    cfg = presets.make_cfg(network='TIMM-vit_giant_patch14_dinov2.lvd142m', num_frames=8, batch_size=1)
    with pytest.raises(NotImplementedError, match='SwiGLU'):
        build_model(cfg, 0)
