"""Small torch building blocks with the parameter names of the classes they stand in for
(so reference checkpoints load unchanged, SURVEY.md section 5 "Checkpoint / resume").  Dense GEMM /
conv / norm work goes to ATen (hipBLASLt / MIOpen) -- host glue around the HIP kernels.
"""
import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn


def _as_fp32(v):
    if isinstance(v, torch.Tensor):
        return v.float() if v.is_floating_point() and v.dtype != torch.float32 else v
    if isinstance(v, (list, tuple)):
        return type(v)(_as_fp32(u) for u in v)
    if isinstance(v, dict):
        return {k: _as_fp32(u) for k, u in v.items()}
    return v


def fp32_region(fn):
    """Decorator for the entry points of boundary B1: the reference evaluates under
    `with autocast():` (train_net.py:334) and keeps only its pixel decoder in fp32 (msdeformattn.py:316).  This build is
    fp32 end to end -- the parity contract is stated against the reference's fp32 CPU path and the HIP operators take
    float32 only -- so every module entry leaves the caller's autocast region, and half-precision feature tensors handed
    in by an autocast caller are up-cast at the edge.  `targets` (the caller-owned prompt memory pool, mutated in place)
    is passed through untouched."""
    import functools
    import inspect
    names = list(inspect.signature(fn).parameters)[1:]          # positional parameter names after `self`

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        if not torch.is_autocast_enabled("cuda"):
            return fn(self, *args, **kwargs)
        with torch.autocast(device_type="cuda", enabled=False):
            args = [a if (i < len(names) and names[i] == "targets") else _as_fp32(a) for i, a in enumerate(args)]
            return fn(self, *args, **{k: (v if k == "targets" else _as_fp32(v)) for k, v in kwargs.items()})
    return wrapper


from .switches import SWITCHES   # split_conv / split_linear: which implementation the layers route to


class Conv2d(nn.Conv2d):
    """detectron2.layers.Conv2d: conv -> optional norm -> optional activation, with the sub-module
    name `norm` (state-dict keys `<name>.weight`, `<name>.norm.weight`, ...)."""

    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def convolve(self, x):
        """the convolution alone (no norm, no activation): for callers that fuse the norm into their next step"""
        y = None
        if (SWITCHES.split_conv and x.is_cuda and self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0)
                and self.dilation == (1, 1) and self.groups == 1):
            from . import ops
            y = ops.conv1x1(x, self.weight, self.bias)
        return y if y is not None else F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)

    def forward(self, x):
        y = None
        if (SWITCHES.split_conv and x.is_cuda and self.bias is None and self.kernel_size == (3, 3) and self.stride == (1, 1)
                and self.padding == (1, 1) and self.dilation == (1, 1) and self.groups == 1):
            from . import ops
            y = ops.conv3x3(x, self.weight)       # None when not covered
        elif (SWITCHES.split_conv and x.is_cuda and self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0)
              and self.dilation == (1, 1) and self.groups == 1):
            from . import ops
            y = ops.conv1x1(x, self.weight, self.bias)   # bias in the epilogue; None when not covered
        x = y if y is not None else F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if isinstance(self.norm, nn.GroupNorm) and x.dtype == torch.float32:
            # GroupNorm (+ ReLU) epilogue through the HIP operator; the module only holds the parameters
            from . import ops
            relu = self.activation is F.relu
            x = ops.group_norm(x, self.norm.num_groups, self.norm.weight, self.norm.bias, self.norm.eps, relu=relu)
            return x if relu or self.activation is None else self.activation(x)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def to_device_async(t, device):
    """Host tensor -> device without blocking the host on the stream: a pageable H2D copy is issued in stream order and waited for, i.e.
    the host stalls until everything enqueued before it has run (the decoder's `frame_indices.to(device)` waited for the whole backbone
    and pixel decoder: 3.7 of the 13 ms a clip took to enqueue, profiles/r04_host_cprofile_step_v1.txt).  Pinned staging + non_blocking."""
    device = torch.device(device)
    if t.device == device:
        return t
    if device.type != "cuda" or t.is_cuda:
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def get_norm(norm, out_channels):
    """detectron2.layers.get_norm for the values the hot path uses ("GN" = GroupNorm(32), "" = none)."""
    if norm is None or norm == "":
        return None
    if norm == "GN":
        return nn.GroupNorm(32, out_channels)
    if norm == "LN":
        return nn.LayerNorm(out_channels)
    raise ValueError(f"unsupported norm {norm!r}")


def point_sample(input, point_coords, **kwargs):
    """detectron2 PointRend `point_sample`: grid_sample on [0,1]^2 coordinates
    (used by the prompt encoder, univs/modeling/prompt_encoder/prompt_encoder.py:127-131)."""
    add_dim = False
    if point_coords.dim() == 3:
        add_dim = True
        point_coords = point_coords.unsqueeze(2)
    output = F.grid_sample(input, 2.0 * point_coords - 1.0, **kwargs)
    if add_dim:
        output = output.squeeze(3)
    return output


class MultiheadAttention(nn.Module):
    """Inference-only multi-head attention with `nn.MultiheadAttention`'s parameter layout
    (`in_proj_weight [3E,E]`, `in_proj_bias`, `out_proj.{weight,bias}`) and calling convention
    (sequence-first `[L, N, E]`, boolean `attn_mask` with True = masked out, shape `[L,S]`,
    `[N*h, L, S]` or -- our extension -- `[N, L, S]` broadcast over heads, which is what the fused
    attention-mask op emits).  Used where the reference uses nn.MultiheadAttention
    (univs/modeling/transformer_decoder/transformer_layers.py:11-148)."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)

    def _packed_rows(self, r0, n):
        """Rows [r0, r0 + n) of the packed in-projection.  Without autograd (inference) a contiguous copy that is made once per weight
        version: a whole tensor whose identity outlives the call, so that the fp16 three-product Linear finds its split image in the cache
        (ops.presplit_weights) instead of splitting the slab of a VIEW again in every launch (13 us per K / V projection of ProCA)."""
        w, b = self.in_proj_weight, self.in_proj_bias
        if torch.is_grad_enabled() and (w.requires_grad or (b is not None and b.requires_grad)):
            return w[r0:r0 + n], (None if b is None else b[r0:r0 + n])
        from . import ops   # (writes through `.data` bump no version counter: ops.invalidate_presplit() is the documented call after them)

        def ver(t):             # (tensors created under torch.inference_mode have no version counter: their identity is the key)
            try:
                return t._version
            except RuntimeError:
                return -1
        key = (r0, n, ver(w), w.data_ptr(), None if b is None else ver(b), ops.presplit_generation())
        cache = self.__dict__.setdefault("_row_cache", {})
        hit = cache.get((r0, n))
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, w[r0:r0 + n].detach().clone(), None if b is None else b[r0:r0 + n].detach().clone())
            cache[(r0, n)] = hit
        return hit[1], hit[2]

    def __getstate__(self):     # the row cache is derived data: neither pickled (torch.save(module)) nor deep-copied
        d = self.__dict__.copy()
        d.pop("_row_cache", None)
        return d

    def forward(self, query, key, value, attn_mask: Optional[torch.Tensor] = None, need_weights=False,
                average_attn_weights=True, kv=None, query_add=None, residual=None, norm=None):
        """`kv` = (k, v) [S, N, E]: the key / value projections when the caller already has them (the decoder computes the K and V
        of all layers that attend to one feature level with ONE Linear each; `key` / `value` are then ignored).
        `query_add`: the query (and, where `key is query`, the key) operand is `query + query_add` -- the layer's `with_pos_embed`
        inside the projection kernel; `residual` + `norm` (an nn.LayerNorm): the first result is `norm(residual + out_proj(.))` --
        the post-norm layer's tail inside the output projection's kernel (transformer_layers.py:42-46, :106-110)."""
        L, N, E = query.shape
        S = key.shape[0] if kv is None else kv[0].shape[0]
        h, d = self.num_heads, self.head_dim
        w, b = self.in_proj_weight, self.in_proj_bias
        small = query.is_cuda and _small(query, w, b)

        def lin(x, r0, n, add=None):
            # rows [r0, r0 + n) of the packed in-projection.  Few rows: one launch with the position add inside (ops.small_linear);
            # tall projections (cross-attention K / V of the 1/8-resolution memory: 73 600 rows) take the fp16 three-product kernel
            if small and x.numel() // E <= 4096:
                from . import ops
                y = ops.small_linear(x, w, b, rows=(r0, n), x_add=add)
                if y is not None:
                    return y
            wr, br = self._packed_rows(r0, n)
            return linear(x if add is None else x + add, wr, br)
        if kv is not None:
            q = lin(query, 0, E, query_add)
            k, v = kv
        elif query is key and key is value and query_add is None:
            q, k, v = lin(query, 0, 3 * E).chunk(3, dim=-1)
        else:
            if query is key:      # self-attention with positional queries (q = k = tgt + pos, v = tgt): q and k in one projection
                qkv = None
                if small and query_add is not None and value is query and E % 32 == 0:
                    from . import ops   # ... and v in the same launch: the position embedding enters the first 2 E features only
                    qkv = ops.small_linear(query, w, b, x_add=query_add, add_features=2 * E)
                if qkv is not None:
                    q, k, v = qkv.chunk(3, dim=-1)
                else:
                    q, k = lin(query, 0, 2 * E, query_add).chunk(2, dim=-1)
                    v = lin(value, 2 * E, E)
            else:
                q = lin(query, 0, E, query_add)
                if key is value:
                    k, v = lin(key, E, 2 * E).chunk(2, dim=-1)
                else:
                    k = lin(key, E, E)
                    v = lin(value, 2 * E, E)

        def project(out):
            if norm is None:
                y = linear(out, self.out_proj.weight, self.out_proj.bias)
                return y if residual is None else residual + y
            return linear_norm(out, self.out_proj, residual, norm)
        if (SWITCHES.fused_cross_attention and query.is_cuda and not need_weights and d == 32 and S >= 64 and L <= 2048
                and (attn_mask is None or (attn_mask.dtype in (torch.bool, torch.uint8)
                                           and ((attn_mask.dim() == 3 and attn_mask.shape[0] == N) or (attn_mask.dim() == 2 and N == 1))))):
            # scores, mask, softmax and P V in one pass over the keys, the [N h, L, S] scores never exist: the decoder's masked
            # cross-attention over the H_l W_l pixels of a level, and its spatio-temporal self-attention over the Q' T query tokens
            # (one batch entry, a [L, S] mask)
            from . import ops
            m3 = attn_mask
            if isinstance(attn_mask, torch.Tensor):
                if S % 4:
                    m3 = ops.pad4_mask(attn_mask)                # (cached per mask object: one padded copy for all layers)
                if m3.dim() == 2:
                    m3 = m3.view(1, L, m3.shape[-1])
            out = ops.cross_attention(q, k, v, m3, h, 1.0 / math.sqrt(d))
            if out is not None:
                return project(out), None
        if attn_mask is not None and not isinstance(attn_mask, torch.Tensor):
            attn_mask = attn_mask.materialize()                  # ops.DeferredMask: the paths below read the reference's tensor
        # [L, N, h, d] -> [N, h, L, d]
        q = q.reshape(L, N, h, d).permute(1, 2, 0, 3)
        k = k.reshape(S, N, h, d).permute(1, 2, 0, 3)
        v = v.reshape(S, N, h, d).permute(1, 2, 0, 3)
        scores = torch.matmul(q * (1.0 / math.sqrt(d)), k.transpose(-1, -2))  # [N, h, L, S]
        fused = attn_mask is None or (attn_mask.dtype in (torch.bool, torch.uint8)
                                      and (attn_mask.dim() == 2 or attn_mask.shape[0] == N))
        if fused:
            # mask + softmax in one in-place pass (HIP operator): per-frame masks [N, L, S] shared by the heads,
            # or one [L, S] mask for all batch entries
            from . import ops
            if attn_mask is not None and attn_mask.dim() == 2:
                attn = ops.masked_softmax_(scores.view(1, N * h, L, S), attn_mask.view(1, L, S)).view(N, h, L, S)
            else:
                attn = ops.masked_softmax_(scores, attn_mask)
        else:
            m = attn_mask.view(N, h, L, S) if attn_mask.shape[0] == N * h else attn_mask.view(N, 1, L, S)
            scores = scores.masked_fill(m, float("-inf")) if m.dtype == torch.bool else scores + m
            attn = torch.softmax(scores, dim=-1)
        out = torch.matmul(attn, v)  # [N, h, L, d]
        out = out.permute(2, 0, 1, 3).reshape(L, N, E)
        out = project(out)
        if need_weights:
            return out, (attn.mean(dim=1) if average_attn_weights else attn)
        return out, None


class MLP(nn.Module):
    """transformer_layers.py:205-217"""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x, transpose01=False, in_norm=None, want_normed=False):
        """`transpose01`: x is [A, B, C]; the result comes back as [B, A, C'] -- contiguous where the last Linear's kernel writes it that
        way (few rows on the GPU), a transposed view otherwise.  `in_norm` (an nn.LayerNorm): the MLP is applied to `in_norm(x)`; with
        `want_normed` the call returns (y, in_norm(x)).  Few rows of 256 channels on the GPU: the whole chain -- LayerNorm included -- is
        ONE launch (ops.small_mlp)."""
        if (SWITCHES.small_mlp_chain and x.is_cuda and self.num_layers <= 3 and x.shape[-1] == 256 and not torch.is_grad_enabled()
                and x.numel() // 256 <= 4096 and (in_norm is None or SWITCHES.small_mlp_norm)):
            from . import ops
            chain = [(l_.weight, l_.bias, i < self.num_layers - 1) for i, l_ in enumerate(self.layers)]
            r = ops.small_mlp(x, chain, in_ln=None if in_norm is None else (in_norm.weight, in_norm.bias, in_norm.eps),
                              want_normed=want_normed, transpose01=transpose01)
            if r is not None:
                return r
        xn = None
        if in_norm is not None:
            x = xn = in_norm(x)
        y = self._forward_layers(x, transpose01)
        return (y, xn) if want_normed else y

    def _forward_layers(self, x, transpose01):
        for i, layer in enumerate(self.layers):
            # hidden layers: the ReLU rides in the GEMM epilogue on the GPU (linear_act: one launch instead of GEMM + bias + clamp)
            if i < self.num_layers - 1:
                x = linear_act(x, layer, F.relu)
            elif transpose01 and x.is_cuda and x.dim() == 3 and _small(x, layer.weight, layer.bias):
                from . import ops
                y = ops.small_linear(x, layer.weight, layer.bias, transpose01=True)
                return y if y is not None else linear(x, layer.weight, layer.bias).transpose(0, 1)
            else:
                x = linear(x, layer.weight, layer.bias)
        return x.transpose(0, 1) if transpose01 else x


def layer_norm(norm, x, residual=None, return_sum=False, post_add=None):
    """`norm(x)` / `norm(x + residual)` for an nn.LayerNorm module through the HIP operator (the module
    only holds the parameters, so the state-dict layout is the reference's); `post_add`: see ops.layer_norm."""
    from . import ops
    return ops.layer_norm(x, norm.weight, norm.bias, norm.eps, residual=residual, return_sum=return_sum, post_add=post_add)


# SWITCHES.split_linear: False = library GEMMs only, True (default) = the MSDeformAttn token projections + encoder FFN
# through the hand-written fp32-accurate GEMM kernels (ops.linear_fused: three fp16 products, csrc/linear_f16x3.hip)


_LIBRARY_FALLBACKS = set()


LIBRARY_LINEAR_COUNTS = {}      # (what, K, N) -> calls that ran on the library GEMM since `reset_library_linear_counts()`: tests assert
                                # per config which Linears may leave the hand-written kernels


def reset_library_linear_counts():
    LIBRARY_LINEAR_COUNTS.clear()


def _note_library_linear(x, weight, what="F.linear"):
    """Count, and log ONCE per (K, N, rows bucket), that a GPU Linear left the hand-written three-product kernels for the library's
    plain-fp32 GEMM: the arithmetic differs in the last bits from the tested path (an odd checkpoint width must not
    change the numerics silently).  The logged reason is the actual one: autograd on, a dtype other than float32, a switch off, or
    the shape itself.  `logging.getLogger("univs_amd")`, level INFO."""
    if not x.is_cuda:
        return
    rows = x.numel() // max(int(x.shape[-1]), 1)
    ck = (what, int(x.shape[-1]), int(weight.shape[0]))
    LIBRARY_LINEAR_COUNTS[ck] = LIBRARY_LINEAR_COUNTS.get(ck, 0) + 1
    key = (what, int(x.shape[-1]), int(weight.shape[0]), rows.bit_length())
    if key in _LIBRARY_FALLBACKS:
        return
    _LIBRARY_FALLBACKS.add(key)
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        why = "autograd is on (the hand-written kernels are inference-only)"
    elif x.dtype != torch.float32 or weight.dtype != torch.float32:
        why = f"dtype {x.dtype} / {weight.dtype} (the hand-written kernels take float32)"
    elif not SWITCHES.split_linear:
        why = "SWITCHES.split_linear is off"
    else:
        why = "shape not covered by the three-product kernels"
    import logging
    logging.getLogger("univs_amd").info("Linear %d -> %d on %d rows runs on the library GEMM (%s): %s", key[1], key[2], rows, what, why)


def linear(x, weight, bias=None):
    """F.linear; on the GPU, tall fp32 projections with K % 128 == 0 or K % 96 == 0 (the MSDeformAttn token projections:
    96 600 rows x 256 -> 256 / 288) take the split kernels on the matrix cores (fp32-accurate: ops.linear_fused),
    everything else ATen."""
    if x.is_cuda and _small(x, weight, bias):
        from . import ops
        y = ops.small_linear(x, weight, bias)
        if y is not None:
            return y
    if SWITCHES.split_linear and x.is_cuda:
        from . import ops
        y = ops.linear_split(x, weight, bias)
        if y is not None:
            return y
    _note_library_linear(x, weight)
    return F.linear(x, weight, bias)


def _small(x, weight, bias=None):
    """the few-rows Linear kernel (ops.small_linear) applies: whole weight tensors only (its split is cached per tensor object --
    slices of a parameter are passed as (parameter, rows) by the callers that have them)"""
    return (SWITCHES.small_linear and x.dtype == torch.float32 and weight._base is None and (bias is None or bias._base is None)
            and x.numel() // max(x.shape[-1], 1) <= 4096 and not torch.is_grad_enabled())


def linear_norm(x, lin, residual, norm):
    """norm(residual + lin(x)) for nn.Linear / nn.LayerNorm modules: one launch for few rows (ops.small_linear), else the Linear and
    the residual LayerNorm kernel"""
    if x.is_cuda and _small(x, lin.weight, lin.bias):
        from . import ops
        y = ops.small_linear(x, lin.weight, lin.bias, residual=residual, ln=(norm.weight, norm.bias, norm.eps))
        if y is not None:
            return y
    return layer_norm(norm, linear(x, lin.weight, lin.bias), residual=residual)


def linear_act(x, linear, activation):
    """activation(linear(x)).  For ReLU on the GPU the activation rides in the GEMM epilogue (hipBLASLt via
    ATen's `_addmm_activation`: bit-identical to relu(linear(x)), one pass over the [tokens, d_ffn]
    activations less -- 0.17 ms per encoder layer at 720p)."""
    if activation is F.relu and x.is_cuda and x.dtype == torch.float32 and linear.bias is not None:
        if _small(x, linear.weight, linear.bias):
            from . import ops
            y = ops.small_linear(x, linear.weight, linear.bias, relu=True)
            if y is not None:
                return y
        if SWITCHES.split_linear:
            from . import ops
            y = ops.linear_split(x, linear.weight, linear.bias, relu=True)
            if y is not None:
                return y
        _note_library_linear(x, linear.weight, "addmm + ReLU")
        y = torch._addmm_activation(linear.bias, x.reshape(-1, x.shape[-1]), linear.weight.t(), use_gelu=False)
        return y.view(*x.shape[:-1], -1)
    return activation(linear(x))


def get_activation_fn(activation):
    if activation == "relu":
        return F.relu
    if activation == "gelu":
        return F.gelu
    if activation == "glu":
        return F.glu
    raise RuntimeError(f"activation should be relu/gelu, not {activation}.")
