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


def mask_decode_attn(mask_embed, feat_lowres, deferred=False):
    # (`deferred`: the HIP operator may hand the all-masked-row rule to its consumer; the stand-in always applies it here)
    logits = torch.einsum("tqc,tchw->tqhw", mask_embed, feat_lowres).flatten(2)
    m = logits.sigmoid() < 0.5
    m[torch.where(m.sum(-1) == m.shape[-1])] = False
    return m


def window_attention(qkv, bias, shift_mask, num_windows, scale, mma="f32"):
    """swin.py:137-168 between the two Linears.  mma="f16" restates UNIVS_MMA_F16 (include/univs_hip.h): q * scale, k, v
    and the un-normalised probabilities rounded to fp16 as operands of fp32-accumulating products, everything else
    fp32 (scores in the exp2 domain, normalisation applied to the output, as the kernel orders it)."""
    B_, N, _, nH, hd = qkv.shape
    q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)  # each [B_, nH, N, hd]
    if mma == "f16":
        log2e = 1.4426950408889634
        r16 = lambda t: t.half().float()            # noqa: E731
        attn = r16(q * (torch.tensor(scale, dtype=torch.float32) * torch.tensor(log2e, dtype=torch.float32))) @ r16(k).transpose(-2, -1) + (bias * log2e).unsqueeze(0)
        if shift_mask is not None:
            nW = shift_mask.shape[0]
            attn = (attn.view(B_ // nW, nW, nH, N, N) + (shift_mask * log2e).unsqueeze(1).unsqueeze(0)).view(-1, nH, N, N)
        e = torch.exp2(attn - attn.amax(-1, keepdim=True))
        out = (r16(e) @ r16(v)) / e.sum(-1, keepdim=True)
        return out.transpose(1, 2).reshape(B_, N, nH * hd)
    assert mma in ("f32", "f16x3"), mma      # "f16x3" is fp32-accurate (three fp16 products per fp32 product): the exact restatement
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


def layer_norm(x, weight, bias, eps=1e-5, residual=None, return_sum=False, post_add=None):
    s = x if residual is None else x + residual
    out = torch.nn.functional.layer_norm(s, (x.shape[-1],), weight, bias, eps)
    if post_add is not None:
        return out, out + post_add
    return (s, out) if return_sum else out


def group_norm(x, num_groups, weight, bias, eps=1e-5, relu=False):
    y = torch.nn.functional.group_norm(x, num_groups, weight, bias, eps)
    return torch.relu(y) if relu else y


def masked_softmax_(scores, mask=None):
    if mask is not None:
        scores.masked_fill_(mask.bool().unsqueeze(1), float("-inf"))
    scores.copy_(torch.softmax(scores, dim=-1))
    return scores


def window_attention_image(qkv, qkv_bias, bias, shift_mask, H, W, window_size, shift, scale, mma="f32"):
    """The reference's data movement around the core (swin.py:252-284): pad (padded pixels carry the qkv bias =
    Linear(0)), roll, window_partition -> core -> window_reverse, roll back, crop."""
    B, L, _, nH, hd = qkv.shape
    ws, F = int(window_size), torch.nn.functional
    x = qkv.reshape(B, H, W, 3 * nH * hd)
    pr, pb = (ws - W % ws) % ws, (ws - H % ws) % ws
    if pr or pb:
        fill = qkv_bias.reshape(-1) if qkv_bias is not None else x.new_zeros(3 * nH * hd)
        xp = fill.view(1, 1, 1, -1).expand(B, H + pb, W + pr, -1).clone()
        xp[:, :H, :W] = x
        x = xp
    Hp, Wp = x.shape[1:3]
    if shift:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = x.view(B, Hp // ws, ws, Wp // ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, 3, nH, hd)
    nW = (Hp // ws) * (Wp // ws)
    o = window_attention(xw, bias, shift_mask if shift else None, nW, scale, mma)         # [B*nW, ws*ws, C]
    o = o.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    if shift:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return o[:, :H, :W].reshape(B, H * W, -1)


def msda_prepare(proj, n_off, reference_points, spatial_shapes, num_heads, num_levels, num_points):
    """ms_deform_attn.py:100-113, the reference's own expressions."""
    N, Lq, _ = proj.shape
    M, L, P = num_heads, num_levels, num_points
    off = proj[..., :M * L * P * 2].reshape(N, Lq, M, L, P, 2)
    attn = torch.softmax(proj[..., n_off:n_off + M * L * P].reshape(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    shapes = spatial_shapes.tolist() if isinstance(spatial_shapes, torch.Tensor) else spatial_shapes
    normalizer = torch.tensor([[w, h] for (h, w) in shapes], dtype=proj.dtype)
    loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    return loc, attn


def msda_forward_fused(value, proj, n_off, reference_points, spatial_shapes, level_start_index, num_points):
    """The fused operator = the reference's own sequence (ms_deform_attn.py:100-116): prepare, then the core."""
    N, S, M, D = value.shape
    shapes = spatial_shapes.tolist() if isinstance(spatial_shapes, torch.Tensor) else spatial_shapes
    L = len(shapes)
    loc, attn = msda_prepare(proj, n_off, reference_points, shapes, M, L, num_points)
    return ms_deform_attn_forward(value, shapes, level_start_index, loc.contiguous(), attn.contiguous())


def msda_set_impl(impl):
    return None


def msda_last_impl():
    return 0


_NAMES = ("ms_deform_attn_forward", "mask_decode", "mask_decode_attn", "window_attention", "bilinear_resample", "layer_norm", "group_norm", "masked_softmax_", "window_attention_image", "msda_prepare")


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
