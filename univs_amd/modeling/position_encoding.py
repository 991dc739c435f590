"""Sine position embeddings of the hot path, written in closed form.

Same values as the reference classes (up to fp32 rounding of identical formulas):
  PositionEmbeddingSine               mask2former/modeling/transformer_decoder/position_encoding.py:12-52
  PositionEmbeddingSine3D ("FixedT")  univs/modeling/transformer_decoder/position_encoding.py:12-110
  PositionEmbeddingSine3DArbitraryT   univs/modeling/transformer_decoder/position_encoding.py:113-236
The reference builds them from cumulative sums of an all-True mask on every call; without padding
masks the cumulative sums are just 1..H / 1..W / 1..T, so the embeddings are pure functions of the
shape (and of the absolute frame indices for ArbitraryT) -- callers cache them per shape.
Channel layout: for a coordinate v, channel 2i = sin(v / T^(2i/F)), channel 2i+1 = cos(v / T^(2i/F)).
"""
import math

import torch
from torch import nn

from ..layers import to_device_async


_DIM_T = {}      # (num_feats, temperature, device) -> the frequency vector: a constant (five tiny launches per call otherwise, and the
                 # prompt path asks for it a dozen times per clip)


def _dim_t(num_feats, temperature, device):
    key = (int(num_feats), float(temperature), str(device))
    v = _DIM_T.get(key)
    if v is None:
        i = torch.arange(num_feats, dtype=torch.float32, device=device)
        v = _DIM_T[key] = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / num_feats)
    return v


def _interleaved_sincos(v, dim_t):
    """v [...], dim_t [F] -> [..., F] with sin on even channels and cos on odd channels."""
    a = v[..., None] / dim_t
    return torch.stack((a[..., 0::2].sin(), a[..., 1::2].cos()), dim=-1).flatten(-2)


class _ShapeCache:
    """The embeddings are pure functions of the shape: one tensor per (shape, device), at most `cap` of them (callers only
    read them: they add them to features or slice views).  Skipped while autograd records."""

    def __init__(self, cap=8):
        self.cap, self.d = cap, {}

    def get(self, key, make):
        if torch.is_grad_enabled():
            return make()
        v = self.d.get(key)
        if v is None:
            if len(self.d) >= self.cap:
                self.d.pop(next(iter(self.d)))
            v = self.d[key] = make()
        return v


def _axis(n, scale, device, normalize=True):
    """cumsum of ones (1..n), normalised like the reference: k / (n + eps) * scale."""
    k = torch.arange(1, n + 1, dtype=torch.float32, device=device)
    if normalize:
        k = k / (float(n) + 1e-6) * scale
    return k


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale
        self._cache = _ShapeCache()

    def forward(self, x, mask=None):
        assert mask is None, "padding masks are not used on the inference hot path"
        N, _, H, W = x.shape
        return self._cache.get((N, H, W, str(x.device)), lambda: self._make(N, H, W, x.device))

    def _make(self, N, H, W, dev):
        dim_t = _dim_t(self.num_pos_feats, self.temperature, dev)
        pos_y = _interleaved_sincos(_axis(H, self.scale, dev, self.normalize), dim_t)  # [H, F]
        pos_x = _interleaved_sincos(_axis(W, self.scale, dev, self.normalize), dim_t)  # [W, F]
        pos = torch.cat((pos_y[:, None, :].expand(H, W, -1), pos_x[None, :, :].expand(H, W, -1)), dim=2)
        return pos.permute(2, 0, 1)[None].expand(N, -1, -1, -1).contiguous()


class _Sine3DBase(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale

    def _yx(self, h, w, device):
        """The spatial part [h, w, 2F] = cat(sin/cos(y), sin/cos(x)) and the frequency table of the temporal part: pure
        functions of the shape (a dozen tiny launches per level otherwise), cached per (h, w, device)."""
        if not hasattr(self, "_yx_cache"):
            self._yx_cache = _ShapeCache(cap=16)

        def make():
            F_ = self.num_pos_feats
            dim_t = _dim_t(F_, self.temperature, device)
            pos_y = _interleaved_sincos(_axis(h, self.scale, device), dim_t)     # [h, F]
            pos_x = _interleaved_sincos(_axis(w, self.scale, device), dim_t)     # [w, F]
            yx = torch.cat((pos_y[:, None, :].expand(h, w, -1), pos_x[None, :, :].expand(h, w, -1)), dim=2).contiguous()
            return yx, _dim_t(2 * F_, self.temperature, device)
        return self._yx_cache.get((h, w, str(device)), make)

    def separable(self, z, h, w, device):
        """(yx [h*w, 2F], pos_z [t, 2F]) with pos[t, :, y, x] = yx[y*w + x] + pos_z[t]: what `_compose` adds up -- for consumers
        that add the embedding to something else anyway (ops.decoder_memory)."""
        yx, dim_t_z = self._yx(h, w, device)
        return yx.view(h * w, -1), _interleaved_sincos(z, dim_t_z)

    def _compose(self, z, h, w, device):
        """z [..t] already scaled -> [..t, 2F, h, w] = cat(sin/cos(y), sin/cos(x)) + sin/cos(z)."""
        yx, dim_t_z = self._yx(h, w, device)
        pos_z = _interleaved_sincos(z, dim_t_z)   # [..t, 2F]
        pos = yx + pos_z[..., None, None, :]       # [..t, h, w, 2F]
        return pos.movedim(-1, -3)                 # [..t, 2F, h, w]

    def _points(self, z, xy):
        """z [t] scaled, xy [n,2] normalised -> [t, n, 2F]."""
        dev = xy.device
        F_ = self.num_pos_feats
        dim_t = _dim_t(F_, self.temperature, dev)
        dim_t_z = _dim_t(2 * F_, self.temperature, dev)
        pos_xy = _interleaved_sincos(xy * self.scale, dim_t)    # [n, 2 (x, y), F]: both coordinates in one pass
        pos_z = _interleaved_sincos(z, dim_t_z)                 # [t, 2F]
        return torch.cat((pos_xy[..., 1, :], pos_xy[..., 0, :]), dim=-1)[None] + pos_z[:, None, :]


class PositionEmbeddingSine3D(_Sine3DBase):
    """"FixedT": z is the frame's position inside the clip (1..T)/T."""

    def forward(self, x, mask=None):
        assert x.dim() == 5 and mask is None
        b, t, _, h, w = x.shape
        dev = x.device
        assert self.normalize
        if not hasattr(self, "_cache"):
            self._cache = _ShapeCache()
        pos = self._cache.get((t, h, w, str(dev)), lambda: self._compose(_axis(t, self.scale, dev), h, w, dev))
        return pos[None].expand(b, -1, -1, -1, -1)

    def forward_separable(self, x):
        """(yx [h*w, 2F], pos_t [1, t, 2F]) whose broadcast sum is `forward(x)` (channels last)."""
        _, t, _, h, w = x.shape
        yx, pz = self.separable(_axis(t, self.scale, x.device), h, w, x.device)
        return yx, pz[None]

    def forward_points_with_size(self, size, xy_embed_normalized):
        t, h, w = size
        assert self.normalize
        z = _axis(t, self.scale, xy_embed_normalized.device)
        return self._points(z, xy_embed_normalized)


class PositionEmbeddingSine3DArbitraryT(_Sine3DBase):
    """z is the ABSOLUTE frame index / num_max_frames (position_encoding.py:123,158)."""

    def __init__(self, num_pos_feats=64, num_max_frames=128, temperature=10000, normalize=False, scale=None):
        super().__init__(num_pos_feats, temperature, normalize, scale)
        assert normalize, "Must enable normalization!"
        self.num_max_frames = num_max_frames

    def _z(self, b, t, dev, t_indices):
        if t_indices is None:
            t_indices = torch.arange(t, device=dev)[None, :].repeat(b, 1)
        return to_device_async(t_indices, dev) / self.num_max_frames * self.scale  # [b, t]

    def temporal(self, t_indices, dev):
        """pos_t [b, t, 2F] = sin / cos of the scaled frame indices: the temporal summand of `forward` / `forward_separable`, the same
        for every feature level of a clip.  For HOST frame indices (what the clip loops pass: `torch.arange(i, i + T)`) the result is
        cached by value -- a dozen tiny launches per level and clip otherwise (position_encoding.py:142-169 recomputes it per call)."""
        def make():
            dim_t_z = _dim_t(2 * self.num_pos_feats, self.temperature, dev)
            z = to_device_async(t_indices, dev) / self.num_max_frames * self.scale
            return _interleaved_sincos(z, dim_t_z)
        if t_indices.is_cuda or torch.is_grad_enabled():
            return make()
        if not hasattr(self, "_t_cache"):
            self._t_cache = _ShapeCache(cap=64)
        key = (tuple(t_indices.shape), tuple(int(v) for v in t_indices.reshape(-1).tolist()), str(dev))
        return self._t_cache.get(key, make)

    def forward(self, x, t_indices=None, mask=None, pos_t=None):
        """`pos_t`: the result of `temporal(t_indices, device)` where the caller already has it."""
        assert x.dim() == 5 and mask is None
        b, t, _, h, w = x.shape
        dev = x.device
        if pos_t is not None:
            yx, _ = self._yx(h, w, dev)
            return (yx + pos_t[..., None, None, :]).movedim(-1, -3)
        return self._compose(self._z(b, t, dev, t_indices), h, w, dev)

    def forward_separable(self, x, t_indices=None, pos_t=None):
        """(yx [h*w, 2F], pos_t [b, t, 2F]) whose broadcast sum is `forward(x, t_indices)` (channels last).  `pos_t`: the result of
        `temporal(t_indices, device)` where the caller already has it (one evaluation for all levels of a clip)."""
        b, t, _, h, w = x.shape
        if pos_t is not None:
            yx, _ = self._yx(h, w, x.device)
            return yx.view(h * w, -1), pos_t
        return self.separable(self._z(b, t, x.device, t_indices), h, w, x.device)

    def forward_points_with_size(self, size, xy_embed_normalized, t_indices=None):
        dev = xy_embed_normalized.device
        t, h, w = size
        if t_indices is None:
            t_indices = torch.arange(t, device=dev)
        if not isinstance(t_indices, torch.Tensor):
            t_indices = torch.as_tensor(t_indices, device=dev)
        assert t_indices.nelement() == 1 or t_indices.nelement() == t, "Unvalid length for frame indices"
        if t_indices.nelement() == 1:
            t_indices = t_indices.reshape(1).repeat(t)
        z = to_device_async(t_indices, dev).reshape(t) / self.num_max_frames * self.scale
        return self._points(z, xy_embed_normalized)
