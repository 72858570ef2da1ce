import os, sys, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests')); sys.path.insert(0, os.path.join(ROOT,'tests','golden'))
import test_gpu_model as T
DEV='cuda'
kw = dict(network='TIMM-vit_base_patch16_224.dino', image_size=224, dropout=0.0, num_frames=32, batch_size=4)
cfg, model = T.make(17, compute_dtype='fp32', **kw)
videos, seq_lens, steps, masks = T.batch(cfg, 18, pad=5)
b, t = 4, 32
x = videos.view(b*2, t, 3, 224, 224).to(DEV)
m2 = masks.view(b*2, 1, t).to(DEV)
model.eval()
out = {}
taps = {}
for mode in ('fp32', 'bf16', 'fp16'):
    model.compute_dtype = mode
    with torch.no_grad():
        out[mode] = model(x, t, video_masks=m2).float().cpu()
        tp, cls = model.features(x)
        taps[mode] = [tt.float().cpu() for tt in tp.tensors]
ref = out['fp32']
for mode in ('bf16', 'fp16'):
    d = (out[mode]-ref).abs()
    print(mode, 'emb max-rel', (d.max()/ref.abs().max()).item(), 'per clip max', [round(v,5) for v in d.amax(dim=(1,2)).tolist()])
    fr = d.amax(dim=2)
    print('   worst frames', [(i//t, i%t, round(fr.flatten()[i].item(),5)) for i in fr.flatten().topk(5).indices.tolist()], 'masks', masks.view(b*2,t)[:, -6:].tolist()[:2])
    for j in range(3):
        e = (taps[mode][j]-taps['fp32'][j])
        print('   tap', j, 'rel-L2', (e.norm()/taps['fp32'][j].norm()).item(), 'max abs', e.abs().max().item(), 'ref max', taps['fp32'][j].abs().max().item(), 'nonfinite', (~torch.isfinite(taps[mode][j])).sum().item())
