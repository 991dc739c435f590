"""Embedding association helpers of the clip loop.

Counterparts of the reference's `univs/inference/comm.py`:
  generate_temporal_weights            univs/inference/comm.py:10-23
  match_from_learnable_embds           univs/inference/comm.py:25-63
  check_consistency_with_prev_frames   univs/inference/comm.py:65-95
Same names / arguments / return conventions.  The assignment itself is a Hungarian solve on a
[N_gt, N_pred] matrix of at most a few hundred entries; like the reference it runs in scipy on the
host (the one D2H copy of the association step), everything around it stays on the device.
"""
import math

import torch
from scipy.optimize import linear_sum_assignment


def generate_temporal_weights(num_frames, weights=None, enable_softmax=False, scaler=5.0):
    """Exponentially increasing weights over `num_frames` frames (newest frame heaviest), optionally
    masked by `weights` [..., T] and normalised to sum 1 (denominator clamped at 1e-3)."""
    w = (torch.arange(1, num_frames + 1).float() / num_frames * scaler).exp()
    if enable_softmax:
        w = w.softmax(-1)
    if weights is not None:
        assert weights.shape[-1] == num_frames, "one weight per frame expected"
        w = w.to(weights) * weights
    return w / w.sum(-1).unsqueeze(-1).clamp(min=1e-3)


def _unit(x):
    return x / x.norm(dim=-1)[..., None].clamp(min=1e-3)


def match_from_learnable_embds(tgt_embds, cur_embds, return_similarity=False, return_src_indices=False,
                               use_norm=True, thresh=0):
    """tgt_embds [N, T_prev, C] x cur_embds [M, T_clip, C] -> Hungarian assignment (target x current).

    use_norm=True: cosine similarity averaged over the clip, temporally weighted over the previous
    frames; use_norm=False: bi-directional softmax ("quasi-dense") of the scaled dot products."""
    t_prev = tgt_embds.shape[1]
    if use_norm:
        cur_embds, tgt_embds = _unit(cur_embds), _unit(tgt_embds)
    sim = torch.einsum("nvc,mtc->nmvt", tgt_embds, cur_embds).mean(-1)          # [N, M, T_prev]
    if use_norm:
        nonblank = (tgt_embds != 0).any(-1).float()
        tw = generate_temporal_weights(t_prev, weights=nonblank)
        sim = (sim * tw.unsqueeze(1)).sum(-1)
    else:
        sim = sim / math.sqrt(tgt_embds.shape[-1])
        sim = (sim.softmax(1) + sim.softmax(0)).mean(-1) / 2.0
        if thresh > 0:
            sim = torch.where(sim < thresh, torch.zeros_like(sim), sim)
    indices = linear_sum_assignment((1 - sim).cpu())
    matched = sim[indices]
    if not return_src_indices:
        indices = indices[1]
    return (indices, matched) if return_similarity else indices


def check_consistency_with_prev_frames(prev_embds, cur_embds, sim_threshold=0.5, return_similarity=False,
                                       use_norm=True):
    """Row-wise (entity e with itself) similarity between the stored embeddings [N, T_prev, C] and the
    clip's [N, T_clip, C]; `is_consistency` = similarity above the threshold."""
    t_prev = prev_embds.shape[1]
    if use_norm:
        cur, prev = _unit(cur_embds), _unit(prev_embds)
        sim = torch.einsum("nvc,ntc->nvt", prev, cur).mean(-1)
        nonblank = (prev != 0).any(-1).float()
        sim = (sim * generate_temporal_weights(t_prev, weights=nonblank)).sum(-1)
        ok = sim > sim_threshold
    else:
        s = torch.einsum("nc,mc->nm", prev_embds[:, -3:].mean(1), cur_embds.mean(1))
        s = 0.5 * (s.softmax(0) + s.softmax(1))
        ok = s.argmax(-1) == torch.arange(len(s), device=s.device)
        sim = torch.diagonal(s, 0)
        ok = ok | (sim > 0.25)
    return (ok, sim) if return_similarity else ok
