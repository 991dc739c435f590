"""CPU execution of the hot path for parity checks and the bench's `cpu_baseline` leg.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The product (`univs_amd/`) has no CPU implementation:
its four operators raise on CPU tensors.  For checking, this module provides CPU stand-ins for exactly
those four operators -- restating the reference's algorithms with plain PyTorch / the plain-C oracle --
and `cpu_ops()`, a context manager that temporarily substitutes them into `univs_amd.ops` so that the
module graph (whose host logic is pinned against the real reference by tests/golden/) can be evaluated
on the host.  Nothing in `univs_amd/` imports this file.

  ms_deform_attn_forward : oracle/ops_ref.c (plain C, OpenMP over queries)     [cuh:242-304]
  mask_decode            : torch.einsum, the reference's own expression        [...decoder_univs.py:527-528]
  mask_decode_attn       : einsum -> sigmoid < 0.5 -> all-masked-row reset      [:527, :565, :390]
  window_attention       : q*scale @ k^T + bias (+mask) -> softmax -> @ v       [swin.py:137-168]\n  bilinear_resample      : F.interpolate(bilinear, align_corners=False)         [...decoder_univs.py:555-558]
"""
import contextlib

import numpy as np
import torch

from . import c_ops


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step=128):
    def tolist(x):
        return x.detach().cpu().reshape(-1).tolist() if isinstance(x, torch.Tensor) else x
    sh = np.asarray(tolist(spatial_shapes), dtype=np.int64).reshape(-1, 2)
    st = np.asarray(tolist(level_start_index), dtype=np.int64).reshape(-1)
    out = c_ops.msda_forward(value.detach().cpu().numpy(), sh, st, sampling_loc.detach().cpu().numpy(),
                             attn_weight.detach().cpu().numpy())
    return torch.from_numpy(out).to(value.device)


def mask_decode(mask_embed, mask_features):
    return torch.einsum("tqc,tchw->qthw", mask_embed, mask_features).contiguous()


def mask_decode_attn(mask_embed, feat_lowres):
    logits = torch.einsum("tqc,tchw->tqhw", mask_embed, feat_lowres).flatten(2)
    m = logits.sigmoid() < 0.5
    m[torch.where(m.sum(-1) == m.shape[-1])] = False
    return m


def window_attention(qkv, bias, shift_mask, num_windows, scale):
    B_, N, _, nH, hd = qkv.shape
    q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)  # each [B_, nH, N, hd]
    attn = (q * scale) @ k.transpose(-2, -1) + bias.unsqueeze(0)
    if shift_mask is not None:
        nW = shift_mask.shape[0]
        attn = attn.view(B_ // nW, nW, nH, N, N) + shift_mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, nH, N, N)
    attn = attn.softmax(dim=-1)
    return (attn @ v).transpose(1, 2).reshape(B_, N, nH * hd)


def bilinear_resample(x, size, addend=None):
    y = torch.nn.functional.interpolate(x, size=tuple(size), mode="bilinear", align_corners=False)
    return y if addend is None else addend + y


def layer_norm(x, weight, bias, eps=1e-5, residual=None, return_sum=False):
    s = x if residual is None else x + residual
    out = torch.nn.functional.layer_norm(s, (x.shape[-1],), weight, bias, eps)
    return (s, out) if return_sum else out


def group_norm(x, num_groups, weight, bias, eps=1e-5, relu=False):
    y = torch.nn.functional.group_norm(x, num_groups, weight, bias, eps)
    return torch.relu(y) if relu else y


def masked_softmax_(scores, mask=None):
    if mask is not None:
        scores.masked_fill_(mask.bool().unsqueeze(1), float("-inf"))
    scores.copy_(torch.softmax(scores, dim=-1))
    return scores


def msda_set_impl(impl):
    return None


def msda_last_impl():
    return 0


_NAMES = ("ms_deform_attn_forward", "mask_decode", "mask_decode_attn", "window_attention", "bilinear_resample", "layer_norm", "group_norm", "masked_softmax_")


@contextlib.contextmanager
def cpu_ops():
    """Temporarily route `univs_amd.ops.<op>` to the CPU stand-ins above."""
    from univs_amd import ops
    saved = {n: getattr(ops, n) for n in _NAMES}
    try:
        for n in _NAMES:
            setattr(ops, n, globals()[n])
        yield
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
