"""Oracle: sequence-contrastive loss (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Vectorised CPU restatement of `SCL.compute_sequence_loss`
(CARL_MVF/algos/scl.py:52-105, `safe_div` :13-16).  Differentiable plain torch;
PINNED by tests/golden/scl_*.npz generated from the imported reference module.

Row r of the flattened [M = B*V*T] embedding matrix belongs to video
b = r // (V*T), view v = (r // T) % V, frame t = r % T (V == 2).
"""
import torch


def row_meta(batch, views, frames, device=None):
    r = torch.arange(batch * views * frames, device=device)
    return r // (views * frames), (r // frames) % views


def scl_loss(embs, seq_lens, steps, masks, negative_type='single_noself', temperature=0.1,
             label_variance=10.0, positive_type='gauss', return_aux=False):
    """embs [B,V,T,E] float; seq_lens [B,V] int; steps [B,V,T] int; masks [B*V,1,T] float.
    Returns the scalar loss (scl.py:105 `{"loss": loss}`)."""
    b, v, t, e = embs.shape
    m = b * v * t
    x = embs.reshape(m, e)
    s = steps.reshape(m)
    ln = seq_lens.reshape(b, v, 1).expand(b, v, t).reshape(m).float()
    mk = masks.reshape(m)
    pair = mk[:, None] * mk[None, :]                                   # scl.py:59

    logits = x @ x.t() / temperature                                   # :61
    # :62 -- int64 / float32 -> float32, left to right
    dist = torch.abs(s.view(-1, 1) / ln.view(-1, 1) * ln.view(1, -1) - s.view(1, -1))
    dist = dist.masked_fill(pair == 0, 1e6)                            # :63

    vid, view = row_meta(b, v, t, x.device)
    same_vid = vid[:, None] == vid[None, :]
    same_view = view[:, None] == view[None, :]
    weight = torch.ones_like(logits)
    if 'single' in negative_type:                                      # :74-76
        weight = weight * same_vid
    if 'noself' in negative_type:                                      # :77-79
        weight = weight * (~(same_vid & same_view))
    weight = weight.masked_fill(pair == 0, 1e-6)                       # :80

    label = torch.zeros_like(logits)
    if positive_type == 'gauss':                                       # :84-96
        pos = torch.exp(-dist * dist / (2 * label_variance)).to(logits.dtype)
        other = same_vid & ~same_view
        pos = pos * other
        rs = pos.sum(1, keepdim=True)
        label = pos / rs
        label = torch.where(torch.isnan(label), torch.zeros_like(label), label)  # safe_div
    ex = torch.exp(logits)                                             # :98  (no max-subtraction)
    sneg = (weight * ex).sum(1, keepdim=True)                          # :99
    prob = ex / sneg
    prob = torch.where(torch.isnan(prob), torch.zeros_like(prob), prob)
    logp = torch.log(prob + 1e-6)
    # F.kl_div(input=logp, target=label, 'none') = xlogy(label, label) - label*logp
    kl = torch.xlogy(label, label) - label * logp                      # :101
    loss = (kl * pair).sum() / mk.sum()                                # :102-103
    if return_aux:
        return loss, {'logits': logits, 'dist': dist, 'weight': weight, 'label': label, 'sneg': sneg}
    return loss
