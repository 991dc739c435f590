"""Swin Transformer backbone for the MI355X hot path.

Interface and state-dict layout follow the reference's `D2SwinTransformer`
(mask2former/modeling/backbone/swin.py: registered at :686-687, forward :745-760 -> :651-678,
`output_shape` :762-768, `size_divisibility` :770-772); the computation is organised differently:

  * window attention core = ONE fused HIP kernel (`ops.window_attention`, f32 MFMA) reading the qkv
    Linear's output in place -- the reference permutes qkv, materialises [B_, nH, N, N] scores and
    makes five more passes over them (swin.py:138-168);
  * the relative-position bias is gathered to [nH, N, N] once per module (cached), the shifted-window
    mask once per (Hp, Wp, device) (cached) -- the reference rebuilds both on every call
    (swin.py:148-155, :413-440);
  * inference only: DropPath / dropout are identities and are not instantiated.
"""

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ... import ops
from ...layers import fp32_region, layer_norm
from ...registry import BACKBONE_REGISTRY, ShapeSpec


def _to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


from ...switches import SWITCHES   # swin_fused_linear / swin_fused_parts: False = the Swin Linears stay on the library GEMM


def _linear(mod, x):
    y = ops.linear_fused(x, mod.weight, mod.bias) if (SWITCHES.swin_fused_linear and (SWITCHES.swin_fused_parts & 1) and x.is_cuda) else None
    return y if y is not None else mod(x)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, out_features)

    def fused(self, x, residual=None, norm=None, post_norm=None):
        """Both Linears, the GELU, the shortcut add and (with `norm`, an nn.LayerNorm applied to x first) the block's norm2 in
        ONE kernel (csrc/mlp_f16x3.hip): stages with C <= 256.  None when the shape is not covered or the fusion is off.
        `post_norm` (an nn.LayerNorm): returns the pair (y, post_norm(y)) -- the next block's norm1, or the stage's output norm,
        from the same launch."""
        if not (SWITCHES.fused_mlp and SWITCHES.swin_fused_linear and (SWITCHES.swin_fused_parts & 6) == 6 and x.is_cuda
                and x.shape[-1] <= SWITCHES.fused_mlp_max_c):
            return None
        ln = None if norm is None else (norm.weight, norm.bias, norm.eps)
        if post_norm is not None:
            return ops.mlp_fused(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, "gelu", residual=residual, ln=ln,
                                 post_ln=(post_norm.weight, post_norm.bias, post_norm.eps), dual=True)
        return ops.mlp_fused(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, "gelu", residual=residual, ln=ln)

    def forward(self, x, residual=None):
        """fc2(GELU(fc1(x))) (+ residual) (swin.py:35-58; the block's `shortcut + mlp(...)` of :291-293 rides in fc2's
        epilogue).  On the GPU both Linears take the three-product fp16 kernel with the GELU / the residual add fused into the
        store where the shape is covered (ops.linear_fused); the library GEMM + elementwise passes otherwise."""
        y = self.fused(x, residual)
        if y is not None:
            return y
        h = ops.linear_fused(x, self.fc1.weight, self.fc1.bias, act="gelu") if (SWITCHES.swin_fused_linear and (SWITCHES.swin_fused_parts & 2) and x.is_cuda) else None
        if h is None:
            h = self.act(self.fc1(x))
        y = ops.linear_fused(h, self.fc2.weight, self.fc2.bias, residual=residual) if (SWITCHES.swin_fused_linear and (SWITCHES.swin_fused_parts & 4) and x.is_cuda) else None
        if y is None:
            y = self.fc2(h)
            if residual is not None:
                y = residual + y
        return y


def window_partition(x, window_size):
    B, H, W, C = x.shape
    x = x.view(B, H // window_size, window_size, W // window_size, window_size, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, window_size, window_size, C)


def window_reverse(windows, window_size, H, W):
    B = int(windows.shape[0] / (H * W / window_size / window_size))
    x = windows.view(B, H // window_size, W // window_size, window_size, window_size, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


class WindowAttention(nn.Module):
    """swin.py:74-171.  Parameters: relative_position_bias_table, qkv, proj; buffer
    relative_position_index."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None):
        super().__init__()
        self.dim = dim
        self.window_size = window_size  # (Wh, Ww)
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((2 * window_size[0] - 1) * (2 * window_size[1] - 1), num_heads))
        coords_h = torch.arange(window_size[0])
        coords_w = torch.arange(window_size[1])
        coords = torch.stack(torch.meshgrid([coords_h, coords_w], indexing="ij"))
        coords_flatten = torch.flatten(coords, 1)
        rel = coords_flatten[:, :, None] - coords_flatten[:, None, :]
        rel = rel.permute(1, 2, 0).contiguous()
        rel[:, :, 0] += window_size[0] - 1
        rel[:, :, 1] += window_size[1] - 1
        rel[:, :, 0] *= 2 * window_size[1] - 1
        self.register_buffer("relative_position_index", rel.sum(-1))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self._bias_cache = None
        # how the two attention products run (SwinTransformer.set_attention_mma): "f16x3" = fp32-accurate on the fp16 matrix
        # cores (two fp16 parts per operand, three products; windows up to 9 x 9, larger ones take the exact kernel),
        # "f32" = exact f32 MFMA, "f16" = fp16 operands (BASELINE config 5; an explicit choice only)
        self.mma = "f16x3"

    def _bias(self):
        t = self.relative_position_bias_table
        key = (t.data_ptr(), t._version, t.device)
        if self._bias_cache is None or self._bias_cache[0] != key:
            n = self.window_size[0] * self.window_size[1]
            b = t[self.relative_position_index.view(-1)].view(n, n, -1).permute(2, 0, 1).contiguous()
            self._bias_cache = (key, b.detach())
        return self._bias_cache[1]

    def forward_image(self, x, H, W, shift, mask=None, residual=None):
        """x: [B, H*W, C] tokens in image order.  Same result as pad -> roll -> window_partition -> forward ->
        window_reverse -> roll -> crop (the reference block, swin.py:252-284), with all of that data movement
        done by index arithmetic inside the attention kernel (the qkv / proj Linears are per token, so they
        commute with the partition).  `residual` [B, H*W, C]: added to the result."""
        B, L, C = x.shape
        qkv = _linear(self.qkv, x).view(B, L, 3, self.num_heads, C // self.num_heads)
        out = ops.window_attention_image(qkv, self.qkv.bias, self._bias(), mask, H, W, self.window_size[0], shift,
                                         self.scale, mma=self.mma)
        if residual is None:
            return _linear(self.proj, out)
        # the block's `shortcut + attn branch` (swin.py:286) in the proj Linear's epilogue
        y = None
        if SWITCHES.swin_fused_linear and (SWITCHES.swin_fused_parts & 1) and out.is_cuda:
            y = ops.linear_fused(out, self.proj.weight, self.proj.bias, residual=residual)
        return y if y is not None else residual + self.proj(out)

    def forward(self, x, mask=None):
        """x: [num_windows*B, N, C]; mask: [nW, N, N] (0 / -100) or None."""
        B_, N, C = x.shape
        qkv = self.qkv(x).view(B_, N, 3, self.num_heads, C // self.num_heads)
        nW = mask.shape[0] if mask is not None else 1
        out = ops.window_attention(qkv, self._bias(), mask, nW, self.scale)
        return self.proj(out)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4.0, qkv_bias=True,
                 qk_scale=None):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.window_size, self.shift_size, self.mlp_ratio = window_size, shift_size, mlp_ratio
        assert 0 <= self.shift_size < self.window_size
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, _to_2tuple(window_size), num_heads, qkv_bias, qk_scale)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.H = None
        self.W = None

    def forward(self, x, mask_matrix, normed=None, next_norm=None):
        """`normed`: norm1(x) when the caller already has it; `next_norm` (an nn.LayerNorm): the block returns (y, next_norm(y)) --
        (y, None) where the fused MLP kernel that provides it does not apply."""
        B, L, C = x.shape
        H, W = self.H, self.W
        assert L == H * W, "input feature has wrong size"
        ws = self.window_size
        shortcut = x
        if x.is_cuda and SWITCHES.fused_mlp:
            # x = shortcut + attn branch from the proj Linear's epilogue; then x + mlp(norm2(x)) in ONE launch where the fused
            # MLP covers the width (its x tile sits in registers: the LayerNorm costs no pass over memory), else norm2 + Mlp
            x = self.attn.forward_image(normed if normed is not None else layer_norm(self.norm1, x), H, W, self.shift_size,
                                        mask_matrix if self.shift_size > 0 else None, residual=shortcut).reshape(B, H * W, C)
            if next_norm is not None and not torch.is_grad_enabled():
                pair = self.mlp.fused(x, residual=x, norm=self.norm2, post_norm=next_norm)
                if pair is not None:
                    return pair
            y = self.mlp.fused(x, residual=x, norm=self.norm2)
            y = y if y is not None else self.mlp(layer_norm(self.norm2, x), residual=x)
            return y if next_norm is None else (y, None)
        if normed is not None or next_norm is not None:
            raise RuntimeError("SwinTransformerBlock: normed / next_norm belong to the fused GPU path")
        x = self.attn.forward_image(layer_norm(self.norm1, x), H, W, self.shift_size,
                                    mask_matrix if self.shift_size > 0 else None)
        # residual add and norm2 in one pass: x = shortcut + attn branch, h = norm2(x)
        x, h = layer_norm(self.norm2, x.reshape(B, H * W, C), residual=shortcut, return_sum=True)
        return self.mlp(h, residual=x)


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x, H, W):
        B, L, C = x.shape
        assert L == H * W
        x = x.view(B, H, W, C)
        if x.is_cuda and not torch.is_grad_enabled():
            # pad + the four strided slices + concatenation + norm in one pass (csrc/layer_norm.hip: patch_merge_norm_kernel)
            h = ops.patch_merge_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
            if h is not None:
                return _linear(self.reduction, h)
        if (H % 2 == 1) or (W % 2 == 1):
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        x = torch.cat([x[:, 0::2, 0::2, :], x[:, 1::2, 0::2, :], x[:, 0::2, 1::2, :], x[:, 1::2, 1::2, :]], -1)
        x = x.view(B, -1, 4 * C)
        return _linear(self.reduction, layer_norm(self.norm, x))   # 4C -> 2C, no bias: the three-product Linear where covered


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 downsample=None):
        super().__init__()
        self.window_size = window_size
        self.shift_size = window_size // 2
        self.depth = depth
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, num_heads, window_size, 0 if (i % 2 == 0) else window_size // 2,
                                 mlp_ratio, qkv_bias, qk_scale) for i in range(depth)])
        self.downsample = downsample(dim=dim) if downsample is not None else None
        self._mask_cache = {}

    def _shift_mask(self, H, W, device):
        """swin.py:413-440, computed once per (H, W, device)."""
        key = (H, W, str(device))
        m = self._mask_cache.get(key)
        if m is None:
            ws, ss = self.window_size, self.shift_size
            Hp = int(np.ceil(H / ws)) * ws
            Wp = int(np.ceil(W / ws)) * ws
            img_mask = torch.zeros((1, Hp, Wp, 1))
            cnt = 0
            for h in (slice(0, -ws), slice(-ws, -ss), slice(-ss, None)):
                for w in (slice(0, -ws), slice(-ws, -ss), slice(-ss, None)):
                    img_mask[:, h, w, :] = cnt
                    cnt += 1
            mw = window_partition(img_mask, ws).view(-1, ws * ws)
            am = mw.unsqueeze(1) - mw.unsqueeze(2)
            am = am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))
            m = am.contiguous().to(device)
            if len(self._mask_cache) > 8:
                self._mask_cache.clear()
            self._mask_cache[key] = m
        return m

    def forward(self, x, H, W, out_norm=None):
        """`out_norm` (an nn.LayerNorm, GPU inference): also returns out_norm(x_out) as a 7th element (None where the last block's fused
        kernel does not provide it) -- the stage's output norm (swin.py:664-672) from the launch that produced x_out."""
        attn_mask = self._shift_mask(H, W, x.device)
        chain = x.is_cuda and SWITCHES.fused_mlp and not torch.is_grad_enabled()
        normed, x_normed = None, None
        for i, blk in enumerate(self.blocks):
            blk.H, blk.W = H, W
            if chain:
                # a block's fused MLP kernel also writes the NEXT block's norm1 of its output (or the stage's output norm)
                nxt = self.blocks[i + 1].norm1 if i + 1 < len(self.blocks) else out_norm
                if nxt is not None:
                    x, n_ = blk(x, attn_mask, normed=normed, next_norm=nxt)
                else:
                    x, n_ = blk(x, attn_mask, normed=normed), None
                if i + 1 < len(self.blocks):
                    normed = n_
                else:
                    x_normed = n_
            else:
                x = blk(x, attn_mask)
        if out_norm is not None:
            if self.downsample is not None:
                x_down = self.downsample(x, H, W)
                return x, H, W, x_down, (H + 1) // 2, (W + 1) // 2, x_normed
            return x, H, W, x, H, W, x_normed
        if self.downsample is not None:
            x_down = self.downsample(x, H, W)
            return x, H, W, x_down, (H + 1) // 2, (W + 1) // 2
        return x, H, W, x, H, W


class PatchEmbed(nn.Module):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, patch_norm=True):
        super().__init__()
        self.patch_size = _to_2tuple(patch_size)
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = nn.LayerNorm(embed_dim) if patch_norm else None

    def forward(self, x):
        _, _, H, W = x.size()
        if W % self.patch_size[1] != 0:
            x = F.pad(x, (0, self.patch_size[1] - W % self.patch_size[1]))
        if H % self.patch_size[0] != 0:
            x = F.pad(x, (0, 0, 0, self.patch_size[0] - H % self.patch_size[0]))
        x = self.proj(x)
        if self.norm is not None:
            Wh, Ww = x.size(2), x.size(3)
            x = layer_norm(self.norm, ops.transpose_last2(x.flatten(2)))
            x = x.transpose(1, 2).view(-1, self.embed_dim, Wh, Ww)
        return x

    def tokens(self, x):
        """(tokens [B, Wh*Ww, E], Wh, Ww): the same as `forward(x).flatten(2).transpose(1, 2)`; on the GPU, for 3-channel images and
        4 x 4 patches, convolution + bias + token layout + patch norm are ONE kernel (ops.patch_embed4)."""
        if x.is_cuda and self.patch_size == (4, 4) and self.in_chans == 3:
            _, _, H, W = x.size()
            xp = x
            if W % 4 != 0:
                xp = F.pad(xp, (0, 4 - W % 4))
            if H % 4 != 0:
                xp = F.pad(xp, (0, 0, 0, 4 - H % 4))
            ln = None if self.norm is None else (self.norm.weight, self.norm.bias, self.norm.eps)
            t = ops.patch_embed4(xp, self.proj.weight, self.proj.bias, ln)
            if t is not None:
                return t, xp.shape[2] // 4, xp.shape[3] // 4
        y = self.forward(x)
        return y.flatten(2).transpose(1, 2), y.size(2), y.size(3)


class SwinTransformer(nn.Module):
    def __init__(self, pretrain_img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=(2, 2, 6, 2),
                 num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 ape=False, patch_norm=True, out_indices=(0, 1, 2, 3)):
        super().__init__()
        self.pretrain_img_size = pretrain_img_size
        self.num_layers = len(depths)
        self.embed_dim, self.ape, self.patch_norm, self.out_indices = embed_dim, ape, patch_norm, out_indices
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim, patch_norm)
        if self.ape:
            pis, ps = _to_2tuple(pretrain_img_size), _to_2tuple(patch_size)
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, embed_dim, pis[0] // ps[0], pis[1] // ps[1]))
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(int(embed_dim * 2 ** i), depths[i], num_heads[i], window_size,
                                          mlp_ratio, qkv_bias, qk_scale,
                                          PatchMerging if (i < self.num_layers - 1) else None))
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i in out_indices:
            self.add_module(f"norm{i}", nn.LayerNorm(self.num_features[i]))

    @fp32_region
    def forward(self, x):
        if SWITCHES.graphs and x.is_cuda and not self.training:
            # hipGraph replay per input shape (univs_amd/graphs.py): one launch instead of ~250
            g = self.__dict__.get("_graphed")
            if g is None:
                from ...graphs import GraphedCallable
                g = self.__dict__["_graphed"] = GraphedCallable(self._forward)
            return g(x)
        return self._forward(x)

    def set_attention_mma(self, mma):
        """How every block's window-attention products run: "f16x3" (default: fp32-accurate, two fp16 parts per operand),
        "f32" (exact f32 MFMA) or "f16" (fp16 MFMA operands, fp32 accumulation / softmax -- what BASELINE config 5 names; the
        reference gets there through autocast, train_net.py:334).  "f16" is an explicit choice of the caller: nothing
        switches it on by itself."""
        if mma not in ops.MMA_DTYPES:
            raise ValueError(f"set_attention_mma: {mma!r} (one of {sorted(ops.MMA_DTYPES)})")
        for m in self.modules():
            if isinstance(m, WindowAttention):
                m.mma = mma
        self._graphed = None         # captured graphs hold the old kernels
        return self

    def _forward(self, x):
        if self.ape:
            x = self.patch_embed(x)
            Wh, Ww = x.size(2), x.size(3)
            ape = F.interpolate(self.absolute_pos_embed, size=(Wh, Ww), mode="bicubic")
            x = (x + ape).flatten(2).transpose(1, 2)
        else:
            x, Wh, Ww = self.patch_embed.tokens(x)
        outs = {}
        for i in range(self.num_layers):
            if i in self.out_indices and x.is_cuda and not torch.is_grad_enabled():
                x_out, H, W, x, Wh, Ww, x_normed = self.layers[i](x, Wh, Ww, out_norm=getattr(self, f"norm{i}"))
            else:
                (x_out, H, W, x, Wh, Ww), x_normed = self.layers[i](x, Wh, Ww), None
            if i in self.out_indices:
                x_out = x_normed if x_normed is not None else layer_norm(getattr(self, f"norm{i}"), x_out)
                # tokens -> NCHW (swin.py:676-683 permute + contiguous): an LDS tile transpose on the GPU
                outs[f"res{i + 2}"] = ops.transpose_last2(x_out).view(-1, self.num_features[i], H, W)
        return outs


@BACKBONE_REGISTRY.register()
class D2SwinTransformer(SwinTransformer):
    """`D2SwinTransformer(cfg, input_shape)` as in swin.py:686-743."""

    def __init__(self, cfg, input_shape=None):
        s = cfg.MODEL.SWIN
        super().__init__(s.PRETRAIN_IMG_SIZE, s.PATCH_SIZE, 3, s.EMBED_DIM, s.DEPTHS, s.NUM_HEADS,
                         s.WINDOW_SIZE, s.MLP_RATIO, s.QKV_BIAS, s.QK_SCALE, s.APE, s.PATCH_NORM)
        self._out_features = s.OUT_FEATURES
        self._out_feature_strides = {"res2": 4, "res3": 8, "res4": 16, "res5": 32}
        self._out_feature_channels = {f"res{i + 2}": self.num_features[i] for i in range(4)}

    def forward(self, x):
        assert x.dim() == 4, f"SwinTransformer takes an input of shape (N, C, H, W). Got {x.shape} instead!"
        y = super().forward(x)
        return {k: v for k, v in y.items() if k in self._out_features}

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name],
                                stride=self._out_feature_strides[name]) for name in self._out_features}

    @property
    def size_divisibility(self):
        return 32
