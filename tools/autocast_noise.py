"""How far is the REFERENCE's own precision mode from fp32 on the head?  CPU only, no reference code: the fp32 ORACLE head
(oracle/head.py, eval mode, project=False) run once plainly and once under torch.autocast(float16) -- the mode the reference trains and
evaluates in (CARL_MVF/train.py:113, evaluate paths under `torch.cuda.amp.autocast()`): linears / matmuls in fp16 with fp16 results,
LayerNorm / softmax in fp32.  Inputs: the ViT's own taps for random frames (oracle backbone, fp32), fresh head weights.
    python tools/autocast_noise.py [clips]
Reads: the max-rel / rel-L2 distance of the autocast embeddings from the fp32 ones = the noise floor a `within 1e-3 of the reference`
bar has on this path; the device's fp16 / bf16 head modes are measured against the same fp32 oracle in tools/fp16_error_budget.py."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from video_rep_learning_amd.utils import presets  # noqa: E402
from video_rep_learning_amd.models import build_model  # noqa: E402
from oracle import model as OM  # noqa: E402
import test_gpu_model as T  # noqa: E402

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
cfg = presets.make_cfg(network='TIMM-vit_base_patch16_224.dino', num_frames=32, batch_size=1, compute_dtype='fp32', dropout=0.0)
torch.manual_seed(cfg.RNG_SEED)
model = build_model(cfg, 0)
vit_cfg, head_cfg, scl_cfg = T.oracle_cfgs(cfg)
params = T.cpu_params(model)
t = 32
g = torch.Generator().manual_seed(5)
frames = torch.randn(nc * t, 3, 224, 224, generator=g)
masks = torch.ones(nc, 1, t)
with torch.no_grad():
    feat, cls = OM.backbone_features(frames, params, vit_cfg)
    ref = OM.forward_from_backbone(feat, cls, nc, t, params, vit_cfg, head_cfg, masks, project=False, training=False).double()
    for dt in (torch.float16, torch.bfloat16):
        with torch.autocast('cpu', dtype=dt):
            e = OM.forward_from_backbone(feat, cls, nc, t, params, vit_cfg, head_cfg, masks, project=False, training=False)
        e = e.double()
        print('oracle head under torch.autocast(%s) against the fp32 oracle head, %d clips x %d frames: max-rel %.3e   rel-L2 %.3e' % (
            str(dt).split('.')[-1], nc, t, ((e - ref).abs().max() / ref.abs().max()).item(), ((e - ref).norm() / ref.norm()).item()), flush=True)
