"""Embedding extraction for evaluation, the inference caller of the hot path.

Mirrors `get_embeddings_dataset` of CARL_MVF/evaluate.py:27-81: the model is put in eval(), every video (batch size 1,
any length L) is cut into ceil(L / EVAL.FRAMES_PER_BATCH) equal chunks, each chunk goes through
`model(chunk, num_steps)` -- no mask, `project=False`, i.e. `F.normalize(embed(x))` (transformer.py:229-230) with the
positional table interpolated to the training length when the chunk length differs from TRAIN.NUM_FRAMES
(models/utils.py:117-120,138-140) -- and the per-frame embeddings of the frames with a label >= 0 are collected.
The downstream metric code (Kendall's tau, retrieval, classification: CARL_MVF/evaluation/*) is CPU numpy/sklearn and
out of scope; it consumes exactly the dict returned here."""
import math

import torch


@torch.no_grad()
def get_embeddings(cfg, model, video):
    """video [1, L, 3, H, W] on the device -> per-frame embeddings [L, E] (float32, on the CPU like the reference)."""
    assert video.size(0) == 1, 'evaluation runs with batch size 1 (evaluate.py:41)'
    seq_len = video.size(1)
    max_frames = cfg.EVAL.FRAMES_PER_BATCH
    num_contexts = cfg.DATA.NUM_CONTEXTS
    num_batches = int(math.ceil(float(seq_len) / max_frames))
    frames_per_batch = int(math.ceil(float(seq_len) / num_batches))
    embs = []
    for i in range(num_batches):
        curr_idx = i * frames_per_batch
        num_steps = min(seq_len - curr_idx, frames_per_batch)
        steps = torch.arange(curr_idx, curr_idx + num_steps)
        if num_contexts != 1:
            stride = cfg.DATA.CONTEXT_STRIDE
            steps = steps.view(-1, 1) + stride * torch.arange(-(num_contexts - 1), 1).view(1, -1)
        steps = torch.clamp(steps.view(-1), 0, seq_len - 1)
        emb = model(video[:, steps.to(video.device)].contiguous(), num_steps)
        embs.append(emb[0].float().cpu())
    return torch.cat(embs, dim=0)


def get_embeddings_dataset(cfg, model, data_loader, device='cuda'):
    """One pass over an evaluation loader yielding the reference tuple
    (video [1,L,3,H,W], frame_label [1,L], seq_len [1], chosen_steps, video_masks, names)."""
    out = {'embs': [], 'labels': [], 'seq_lens': [], 'input_lens': [], 'steps': [], 'names': []}
    was_training = model.training
    model.eval()
    for video, frame_label, seq_len, chosen_steps, _masks, names in data_loader:
        assert video.size(0) == 1 and video.size(1) == frame_label.size(1) == int(seq_len.item())
        embs = get_embeddings(cfg, model, video.to(device))
        valid = frame_label[0] >= 0
        out['embs'].append(embs[valid.cpu()].numpy())
        out['labels'].append(frame_label[0][valid].cpu().numpy())
        out['seq_lens'].append(int(seq_len.item()))
        out['input_lens'].append(len(video[0]))
        out['steps'].append(chosen_steps[0].cpu().numpy())
        out['names'].append(names[0])
    model.train(was_training)
    return out
