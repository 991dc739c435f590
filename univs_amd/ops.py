"""Tensor-level wrappers over the C ABI (include/univs_hip.h): checks + allocation + launch on the
current torch stream.  PyTorch is plumbing here (device memory, streams); the compute is in
libunivs_hip.so.  Every function raises on non-GPU tensors -- the reference does the same for its
operator ("Not implemented on the CPU", ops/src/ms_deform_attn.h:43) and there is no CPU fallback.
"""
import contextlib
import weakref
import ctypes
import math
import os

import torch

from . import _lib

# Host cost of a wrapper call matters: a clip is ~250 calls into the library, and the decoder's kernels are shorter than a
# Python call.  The three helpers below are the cheap forms of `torch.cuda.current_stream(dev).cuda_stream` (6 us),
# `with torch.cuda.device(dev)` (5-8 us) and `ctypes.c_void_p(t.data_ptr())`: the raw stream of the tensor's device straight
# from the C extension, a device switch only when the tensor does NOT live on the current device, and plain integers
# (ctypes converts them for the declared c_void_p parameters; None stays NULL).
_raw_stream = torch._C._cuda_getCurrentRawStream


def _stream_ptr(t):
    return _raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device())


def _ptr(t):
    return t.data_ptr()


class _on:
    """`with _on(t):` -- the current CUDA device is t's device inside the block (what `torch.cuda.device(t.device)` does), at
    the cost of one integer comparison when it already is."""
    __slots__ = ("idx", "prev")

    def __init__(self, t):
        self.idx = t.device.index
        self.prev = -1

    def __enter__(self):
        if self.idx is not None:
            cur = torch.cuda.current_device()
            if cur != self.idx:
                self.prev = cur
                torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
        return False


def _require_gpu(name, *tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(f"{name}: Not implemented on the CPU (tensor on {t.device}); the HIP "
                               "extension is the only implementation")
        if not t.is_contiguous():
            raise RuntimeError(f"{name}: all tensors have to be contiguous")


def _inference_only(name, *tensors):
    """The kernels behind these wrappers return tensors that are detached from autograd.  Called with gradient
    recording on and an input that requires grad they would train with silently missing gradients, so that
    case raises; wrap inference in `torch.no_grad()` (the reference's eval loop does, train_net.py:334) or use
    the autograd pair `ms_deform_attn_forward` / `ms_deform_attn_backward` through an autograd.Function."""
    if torch.is_grad_enabled():
        for t in tensors:
            if t is not None and t.requires_grad:
                raise RuntimeError(f"{name}: inference-only HIP operator called with gradient recording enabled on a "
                                   "tensor that requires grad; its result would be detached from autograd. Use "
                                   "torch.no_grad() / torch.inference_mode().")


def needs_grad(*tensors):
    """True when autograd would record an op on these tensors (layers.linear then keeps the ATen path)."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _host_shapes(spatial_shapes, level_start_index, S=None):
    """Level table as host int64 ctypes arrays.  Accepts tensors (device or host), lists or tuples.
    A device tensor costs one small D2H copy per call (the reference's operator reads the table on the
    device; there is deliberately NO cache here: an address/version key does not identify the contents
    of a freshly allocated tensor).  The encoder passes the Python lists it already owns, so the hot
    path never syncs.  The table is validated: start[l] must be the running sum of H*W and, when `S`
    is given, the levels must cover exactly S tokens -- an inconsistent table raises instead of sampling
    with the wrong geometry."""
    def to_list(x):
        if isinstance(x, torch.Tensor):
            return [int(v) for v in x.detach().cpu().reshape(-1).tolist()]
        out = []
        for v in x:
            if isinstance(v, (list, tuple)):
                out.extend(int(u) for u in v)
            else:
                out.append(int(v))
        return out
    sh = to_list(spatial_shapes)
    st = to_list(level_start_index)
    L = len(st)
    if len(sh) != 2 * L:
        raise RuntimeError(f"spatial_shapes has {len(sh)} entries, expected 2*{L}")
    if S is not None:
        run = 0
        for l in range(L):
            if sh[2 * l] <= 0 or sh[2 * l + 1] <= 0:
                raise RuntimeError(f"spatial_shapes[{l}] = ({sh[2 * l]}, {sh[2 * l + 1]}) is not positive")
            if st[l] != run:
                raise RuntimeError(f"level_start_index[{l}] = {st[l]} but the levels before it hold {run} tokens")
            run += sh[2 * l] * sh[2 * l + 1]
        if run != int(S):
            raise RuntimeError(f"spatial_shapes cover {run} tokens but value has {int(S)}")
    return (ctypes.c_int64 * len(sh))(*sh), (ctypes.c_int64 * L)(*st), L


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                           im2col_step=128):
    """Same contract as the reference's `MSDA.ms_deform_attn_forward`
    (ops/src/vision.cpp:19; checks from ops/src/cuda/ms_deform_attn_cuda.cu:33-57).
    value [N,S,M,D], sampling_loc [N,Lq,M,L,P,2], attn_weight [N,Lq,M,L,P] -> [N,Lq,M*D]."""
    _require_gpu("ms_deform_attn_forward", value, sampling_loc, attn_weight)
    if value.dtype not in (torch.float32, torch.float64):
        raise RuntimeError(f"ms_deform_attn_forward: unsupported dtype {value.dtype} (float/double only, "
                           "as the reference's AT_DISPATCH_FLOATING_TYPES)")
    if sampling_loc.dtype != value.dtype or attn_weight.dtype != value.dtype:
        raise RuntimeError("ms_deform_attn_forward: value / sampling_loc / attn_weight dtypes differ")
    N, S, M, D = value.shape
    _, Lq, M2, L, P, two = sampling_loc.shape
    if M2 != M or two != 2 or tuple(attn_weight.shape) != (N, Lq, M, L, P):
        raise RuntimeError("ms_deform_attn_forward: inconsistent shapes "
                           f"value={tuple(value.shape)} loc={tuple(sampling_loc.shape)} w={tuple(attn_weight.shape)}")
    step = min(N, int(im2col_step)) if N > 0 else 1
    if step <= 0 or N % step != 0:
        raise RuntimeError(f"batch({N}) must divide im2col_step({step})")
    sh, st, L2 = _host_shapes(spatial_shapes, level_start_index, S)
    if L2 != L:
        raise RuntimeError(f"ms_deform_attn_forward: {L2} levels in spatial_shapes, {L} in sampling_loc")
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    lib = _lib.load()
    fn = lib.univs_msda_forward_f32 if value.dtype == torch.float32 else lib.univs_msda_forward_f64
    with _on(value):
        rc = fn(_ptr(value), sh, st, _ptr(sampling_loc), _ptr(attn_weight), N, S, M, D, L, Lq, P,
                _ptr(out), _stream_ptr(value))
    _lib.check(rc, "ms_deform_attn_forward")
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                            grad_output, im2col_step=128):
    """ops/src/vision.cpp:20 / ms_deform_attn.h:46-66: returns [grad_value, grad_sampling_loc,
    grad_attn_weight] (float32; `im2col_step` accepted and ignored like in the forward)."""
    _require_gpu("ms_deform_attn_backward", value, sampling_loc, attn_weight, grad_output)
    if value.dtype != torch.float32:
        raise RuntimeError("ms_deform_attn_backward: float32 only")
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    shapes, starts, L2 = _host_shapes(spatial_shapes, level_start_index, S)
    if L2 != L or tuple(grad_output.shape) != (N, Lq, M * D) or tuple(attn_weight.shape) != (N, Lq, M, L, P):
        raise RuntimeError("ms_deform_attn_backward: inconsistent shapes")
    gv = torch.empty_like(value)
    gl = torch.empty_like(sampling_loc)
    ga = torch.empty_like(attn_weight)
    with _on(value):
        rc = _lib.load().univs_msda_backward_f32(_ptr(value), shapes, starts, _ptr(sampling_loc), _ptr(attn_weight),
                                                 _ptr(grad_output), N, S, M, D, L, Lq, P, _ptr(gv), _ptr(gl), _ptr(ga),
                                                 _stream_ptr(value))
    _lib.check(rc, "ms_deform_attn_backward")
    return [gv, gl, ga]


def msda_level_order(spatial_shapes):
    """Slot order of the levels in the head-major projection layout of `msda_forward_strips`: by size, largest first, ties
    by index (csrc/msda_strips_geom.h: s5_build_host)."""
    sh = [(int(h), int(w)) for h, w in (spatial_shapes.tolist() if isinstance(spatial_shapes, torch.Tensor) else spatial_shapes)]
    return sorted(range(len(sh)), key=lambda l: (-sh[l][0] * sh[l][1], l))


def msda_pack_head_major(value, proj, n_off, spatial_shapes, num_points=4):
    """Standard layouts -> the head-major operands of `msda_forward_strips`, with torch copies (tests, tools and shapes the
    blocked Linear epilogue does not cover; the hot path gets these layouts from `linear_blocked` for free):
    value [N, S, M, 32] -> [N, M*2, S, 16]; proj [N, S, C] (offsets in columns [0, M*L*P*2), logits from n_off) ->
    [N, M, S, P*3L] with the levels in `msda_level_order`."""
    N, S, M, D = value.shape
    order = msda_level_order(spatial_shapes)
    L, P = len(order), int(num_points)
    off = proj[..., :M * L * P * 2].reshape(N, S, M, L, P, 2)[:, :, :, order]
    lg = proj[..., n_off:n_off + M * L * P].reshape(N, S, M, L, P)[:, :, :, order]
    row = torch.cat([off.permute(0, 2, 1, 4, 3, 5).reshape(N, M, S, P, 2 * L), lg.permute(0, 2, 1, 4, 3)], -1)
    value_hm = value.reshape(N, S, M, 2, D // 2).permute(0, 2, 3, 1, 4).contiguous().view(N, 2 * M, S, D // 2)
    return value_hm, row.reshape(N, M, S, P * 3 * L).contiguous()


def msda_pack_heads(value, proj, n_off, spatial_shapes, num_points=4):
    """Standard layouts -> the head-major operands of `msda_forward_heads` (torch copies: tests and tools; the hot path gets
    them from `linear_blocked`): value [N, S, M, 32] -> [N, M, S, 32]; proj as `msda_pack_head_major`."""
    _, qhm = msda_pack_head_major(value, proj, n_off, spatial_shapes, num_points)
    return value.permute(0, 2, 1, 3).contiguous(), qhm


def msda_forward_heads(value_hm, proj_hm, ref_points, spatial_shapes, level_start_index, num_heads, num_points=4):
    """MSDeformAttn core (ms_deform_attn.py:100-116) on head-major operands, a full head per lane-sample (include/univs_hip.h:
    univs_msda_forward_heads_f32; csrc/msda_heads.hip): value_hm [N, M, S, 32] and proj_hm [N, M, S, P*3L] as `linear_blocked`
    writes them (levels of proj_hm in `msda_level_order`), ref_points [N or 1, S, 2] (one per query, shared by the levels).
    Returns [N, S, M*32], or None when the geometry is not covered."""
    _inference_only("msda_forward_heads", value_hm, proj_hm, ref_points)
    _require_gpu("msda_forward_heads", value_hm, proj_hm, ref_points)
    if any(t.dtype != torch.float32 or not t.is_contiguous() for t in (value_hm, proj_hm, ref_points)):
        raise RuntimeError("msda_forward_heads: contiguous float32 operands only")
    M, P = int(num_heads), int(num_points)
    N, M1, S, D = value_hm.shape
    sh, st, L = _host_shapes(spatial_shapes, level_start_index, S)
    if M1 != M or D != 32 or tuple(proj_hm.shape) != (N, M, S, P * 3 * L) or tuple(ref_points.shape[1:]) != (S, 2) \
            or ref_points.shape[0] not in (1, N):
        raise RuntimeError("msda_forward_heads: inconsistent shapes")
    if P != 4 or not (1 <= L <= 4):
        return None
    out = torch.empty((N, S, M * 32), dtype=torch.float32, device=value_hm.device)
    rbs = 0 if ref_points.shape[0] == 1 else S * 2
    with _on(value_hm):
        rc = _lib.load().univs_msda_forward_heads_f32(_ptr(value_hm), sh, st, _ptr(proj_hm), _ptr(ref_points), rbs, N, S, M, 32,
                                                      L, S, P, _ptr(out), _stream_ptr(value_hm))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "msda_forward_heads")
    return out


def msda_forward_strips(value_hm, proj_hm, ref_points, spatial_shapes, level_start_index, num_heads, num_points=4):
    """MSDeformAttn core (ms_deform_attn.py:100-116) on head-major operands (include/univs_hip.h:
    univs_msda_forward_strips_f32): value_hm [N, M*2, S, 16] and proj_hm [N, M, S, P*3L] as `linear_blocked` writes them
    (levels of proj_hm in `msda_level_order`), ref_points [N or 1, S, 2] (one per query, shared by the levels).
    Returns [N, S, M*32], or None when the geometry is not covered."""
    _inference_only("msda_forward_strips", value_hm, proj_hm, ref_points)
    _require_gpu("msda_forward_strips", value_hm, proj_hm, ref_points)
    if any(t.dtype != torch.float32 or not t.is_contiguous() for t in (value_hm, proj_hm, ref_points)):
        raise RuntimeError("msda_forward_strips: contiguous float32 operands only")
    M, P = int(num_heads), int(num_points)
    N, M2, S, DH = value_hm.shape
    sh, st, L = _host_shapes(spatial_shapes, level_start_index, S)
    if M2 != 2 * M or DH != 16 or tuple(proj_hm.shape) != (N, M, S, P * 3 * L) or tuple(ref_points.shape[1:]) != (S, 2) \
            or ref_points.shape[0] not in (1, N):
        raise RuntimeError("msda_forward_strips: inconsistent shapes")
    if P != 4 or not (1 <= L <= 4):
        return None
    out = torch.empty((N, S, M * 32), dtype=torch.float32, device=value_hm.device)
    rbs = 0 if ref_points.shape[0] == 1 else S * 2
    with _on(value_hm):
        rc = _lib.load().univs_msda_forward_strips_f32(_ptr(value_hm), sh, st, _ptr(proj_hm), _ptr(ref_points), rbs, N, S, M, 32,
                                                       L, S, P, _ptr(out), _stream_ptr(value_hm))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "msda_forward_strips")
    return out


class UnivsConfig(ctypes.Structure):
    """include/univs_hip.h: UnivsConfig -- the library's process-wide settings (it reads no environment variable)."""
    _fields_ = [(n, ctypes.c_int) for n in ("size", "msda_impl", "msda_strip_w", "msda_strip_h", "msda_halo", "msda_grid",
                                            "mask_decode_impl", "mask_decode_ct", "mask_decode_ablate", "window_attn_v1",
                                            "linear_terms", "linear_ablate", "mask_decode_chunked", "mask_decode_wave_tiles",
                                            "linear_rows_per_pass", "linear_grid_x", "xattn_segments", "msda_sched")] + [("reserved", ctypes.c_int * 2)]


def get_config() -> dict:
    c = UnivsConfig()
    _lib.check(_lib.load().univs_get_config(ctypes.byref(c)), "get_config")
    return {n: getattr(c, n) for n, _ in UnivsConfig._fields_ if n not in ("size", "reserved")}


def configure(**settings):
    """Set library settings by name (the others keep their current values); `configure()` with no argument restores the
    defaults.  Returns the previous settings (pass them back to restore)."""
    global _LINEAR_TERMS
    prev = get_config()
    if not settings:
        _lib.check(_lib.load().univs_configure(None), "configure")
        _LINEAR_TERMS = 3
        return prev
    c = UnivsConfig()
    c.size = ctypes.sizeof(UnivsConfig)
    for n, v in {**prev, **settings}.items():
        if n not in prev:
            raise KeyError(f"configure: unknown setting {n!r} (known: {sorted(prev)})")
        setattr(c, n, int(v))
    _lib.check(_lib.load().univs_configure(ctypes.byref(c)), "configure")
    _LINEAR_TERMS = 6 if int(c.linear_terms) == 6 else 3
    return prev


_LINEAR_TERMS = 3        # mirror of UnivsConfig.linear_terms (all changes go through `configure`): the hot path does not query the library


@contextlib.contextmanager
def configured(**settings):
    """`with ops.configured(msda_impl=1): ...` -- settings for the duration of a block (tests, kernel benchmarks)."""
    prev = configure(**settings) if settings else get_config()
    try:
        yield
    finally:
        configure(**prev)


def msda_set_impl(impl: int):
    """0 auto, 1 generic direct-gather kernel, 2 LDS-tiled encoder kernel."""
    _lib.check(_lib.load().univs_msda_set_impl(int(impl)), "msda_set_impl")


def msda_last_impl() -> int:
    """1 = generic kernel, 2 = LDS-tiled kernel ran for the last forward on this thread."""
    return int(_lib.load().univs_msda_last_impl())


def msda_last_tiled_generation() -> int:
    """5 (strips, head-major operands) / 2 = generation of the LDS-tiled kernel that ran for the last forward on this
    thread, 0 = generic."""
    return int(_lib.load().univs_msda_last_tiled_generation())


_ACTS = {None: 0, "none": 0, "relu": 1, "gelu": 2}
# K <= 768: the W-resident kernel (csrc/linear_f16x3.hip); K >= SWITCHES.presplit_kmin: the streamed kernel on weights split once
# per tensor (csrc/gemm_f16x3_stream.hip); anything neither covers goes back to the library GEMM
from .switches import SWITCHES   # linear_kmax: widest K routed to the hand-written Linears


_PRESPLIT = {}      # (id(weight tensor), mode) -> (weak reference to it, (version, data_ptr, device), wp, winv, event recorded behind
                    # the split, ids of the streams already ordered behind it); dropped with the tensor
PRESPLIT_MODES = {"linear": 0, "conv": 1, "mlp2": 2}    # include/univs_hip.h: univs_presplit_weights_f32 `conv`


_PRESPLIT_GENERATION = [0]   # bumped by invalidate_presplit: derived caches (layers.MultiheadAttention._packed_rows) key on it


def presplit_generation():
    return _PRESPLIT_GENERATION[0]


def invalidate_presplit(weight=None):
    """Drop the cached splits of `weight` (all of them without an argument).  The cache notices in-place updates that bump the
    tensor's version counter (load_state_dict, optimizer steps, `weight.copy_`) and new storages; writes through `weight.data`
    (common in checkpoint / EMA code) bump a DIFFERENT counter and leave the address unchanged -- after those, call this."""
    _PRESPLIT_GENERATION[0] += 1
    if weight is None:
        _PRESPLIT.clear()
        return
    for mode in PRESPLIT_MODES.values():
        _PRESPLIT.pop((id(weight), mode), None)


def presplit_weights(weight, conv=False, mode=None):
    """The split of a weight tensor for the three-product fp16 GEMM kernels (include/univs_hip.h:
    univs_presplit_weights_f32): `weight` [N, K] (a Linear) or [N, Cin, 3, 3] with `conv=True` -> (wp [N * K] int32 = two fp16
    parts per element in the kernels' LDS order, winv [N]).  `mode="mlp2"`: the second Linear of `mlp_fused` (its k-order).
    Done once per tensor and mode and cached; an in-place update (load_state_dict, optimizer step), a move to another device
    or a new storage invalidates the entry (writes through `.data`: see `invalidate_presplit`)."""
    mode = PRESPLIT_MODES["conv" if conv else (mode or "linear")]
    key = (weight._version, weight.data_ptr(), weight.device)
    e = _PRESPLIT.get((id(weight), mode))
    if e is not None and e[0]() is weight and e[1] == key:
        # made on another stream (the prompt sampler's side stream, a caller's own): order this stream behind the split, once
        # (the raw handle first: building the Stream object costs ~5 us, and this is the path of ~190 calls per clip)
        sid = _stream_ptr(weight)
        if sid not in e[5] and not torch.cuda.is_current_stream_capturing():   # (a capture follows eager warm-up calls: graphs.py)
            cur = torch.cuda.current_stream(weight.device)
            cur.wait_event(e[4])
            e[2].record_stream(cur)
            e[3].record_stream(cur)
            e[5].add(sid)
        return e[2], e[3]
    _require_gpu("presplit_weights", weight)
    if weight.dtype != torch.float32:
        raise RuntimeError("presplit_weights: float32 only")
    N = weight.shape[0]
    K = weight.numel() // max(N, 1)
    if mode == 1 and (weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3)):
        raise RuntimeError("presplit_weights: conv=True takes a [N, Cin, 3, 3] weight")
    w = weight.detach().contiguous()
    wp = torch.empty(N * K, dtype=torch.int32, device=weight.device)
    winv = torch.empty(N, dtype=torch.float32, device=weight.device)
    with _on(weight):
        _lib.check(_lib.load().univs_presplit_weights_f32(_ptr(w), N, K, mode, _ptr(wp), _ptr(winv), _stream_ptr(w)),
                   "presplit_weights")
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(weight.device))
    wid = (id(weight), mode)
    _PRESPLIT[wid] = (weakref.ref(weight, lambda _r, _i=wid: _PRESPLIT.pop(_i, None)), key, wp, winv, done,
                      {torch.cuda.current_stream(weight.device).cuda_stream})
    return wp, winv


MLP_WIDTHS = (96, 128, 192, 256, 384)


def mlp_fused(x, w1, b1, w2, b2, act, residual=None, ln=None, post_ln=None, post_add=None, residual_normed=False, dual=False):
    """act(LN(x) W1^T + b1) W2^T + b2 (+ residual) in ONE kernel (include/univs_hip.h: univs_mlp_presplit_f32; csrc/mlp_f16x3.hip):
    the encoder FFN (msdeformattn.py:87-91) and the Swin Mlp + shortcut (swin.py:35-58, :291-293).  Both products use the
    three-product fp16 arithmetic of `linear_fused`; the [M, Hd] hidden activations stay in registers.  `act`: 'relu' | 'gelu'.
    `ln` = (weight, bias, eps) of an nn.LayerNorm(C) applied to the rows of x inside the kernel first (the Swin block's
    `x + mlp(norm2(x))`, swin.py:289-293, is then one launch with residual=x).  `post_ln` = (weight, bias, eps): the finished rows
    (bias and residual added) go through that LayerNorm before they are stored -- the encoder layer's `norm2(src + ffn(src))`,
    msdeformattn.py:91-95 -- and with `post_add` [rows, C] (rows dividing M: broadcast over the leading dimension) the call
    returns the pair (y, y + post_add), the second being the next layer's `with_pos_embed(src, pos)`.
    `residual_normed=True` (with `ln`, without `residual`): the residual is LN(x) itself -- `x1 = norm1(x); norm2(x1 + ffn(x1))`, the
    whole tail of the encoder layer behind `src + output_proj(...)` (msdeformattn.py:124-133) in one launch.
    `dual=True` (with `post_ln`, without `post_add`): returns (y, post_ln(y)) with y the finished rows UN-normalised -- a Swin block's
    output and the next block's `norm1` of it (or the stage's output norm) from the same launch.
    Returns None when the shape is not covered (C not in 96 / 128 / 192 / 256 / 384, Hd % 32, fewer than 2048 rows, autograd needed):
    the caller keeps two `linear_fused` calls."""
    C = x.shape[-1]
    Hd = w1.shape[0]
    M = x.numel() // max(C, 1)
    if needs_grad(x, w1, b1, w2, b2, residual) or act not in ("relu", "gelu"):
        return None
    if residual_normed and (ln is None or residual is not None):
        raise RuntimeError("mlp_fused: residual_normed needs ln and excludes residual")
    if dual and (post_ln is None or post_add is not None or residual_normed):
        raise RuntimeError("mlp_fused: dual needs post_ln and excludes post_add / residual_normed")
    if (not x.is_cuda or x.dtype != torch.float32 or w1.dtype != torch.float32 or w2.dtype != torch.float32 or C not in MLP_WIDTHS
            or tuple(w1.shape) != (Hd, C) or tuple(w2.shape) != (C, Hd) or Hd % 32 != 0 or M < 2048 or M * C * 4 >= 2 ** 31 - 1
            or ((1 if C == 384 else 2) * 64 * C + 2 * Hd + 6 * C) * 4 > 160 * 1024):
        return None
    pw = pb = pa = None
    peps, parows = 0.0, 0
    if post_ln is not None:
        pw, pb, peps = post_ln
        for t_ in (pw, pb):
            if t_ is not None and (t_.dtype != torch.float32 or tuple(t_.shape) != (C,) or not t_.is_cuda or not t_.is_contiguous()):
                raise RuntimeError("mlp_fused: LayerNorm weight / bias must be contiguous float32 [C] on the GPU")
        if pw is None:
            raise RuntimeError("mlp_fused: post_ln needs a weight")
    if post_add is not None:
        if post_ln is None:
            raise RuntimeError("mlp_fused: post_add needs post_ln")
        pa = post_add.contiguous()
        parows = pa.numel() // C
        if pa.dtype != torch.float32 or not pa.is_cuda or pa.shape[-1] != C or parows < 1 or M % parows != 0:
            raise RuntimeError("mlp_fused: post_add must be float32 [rows, C] on the GPU with rows dividing the number of tokens")
    lw = lb = None
    leps = 0.0
    if ln is not None:
        lw, lb, leps = ln
        for t_ in (lw, lb):
            if t_ is not None and (t_.dtype != torch.float32 or tuple(t_.shape) != (C,) or not t_.is_cuda or not t_.is_contiguous()):
                raise RuntimeError("mlp_fused: LayerNorm weight / bias must be contiguous float32 [C] on the GPU")
        if lw is None:
            raise RuntimeError("mlp_fused: ln needs a weight")
    x2 = x.contiguous().view(M, C)
    _require_gpu("mlp_fused", x2)
    for b, n in ((b1, Hd), (b2, C)):
        if b is not None and (b.dtype != torch.float32 or tuple(b.shape) != (n,) or not b.is_cuda or not b.is_contiguous()):
            raise RuntimeError("mlp_fused: biases must be contiguous float32 [Hd] / [C] on the GPU")
    r = None
    if residual is not None:
        if residual.dtype != torch.float32 or not residual.is_cuda or tuple(residual.shape) != tuple(x.shape):
            raise RuntimeError(f"mlp_fused: residual must be float32 of x's shape on the GPU (got {tuple(residual.shape)})")
        r = residual.contiguous()
    y = torch.empty((M, C), dtype=torch.float32, device=x.device)
    y2 = torch.empty((M, C), dtype=torch.float32, device=x.device) if (pa is not None or dual) else None
    with _on(x):
        w1p, w1inv = presplit_weights(w1)
        w2p, w2inv = presplit_weights(w2, mode="mlp2")
        rc = _lib.load().univs_mlp_presplit_v2_f32(_ptr(x2), _ptr(w1p), _ptr(w1inv), _ptr(b1) if b1 is not None else None, _ptr(w2p),
                                                _ptr(w2inv), _ptr(b2) if b2 is not None else None, _ptr(r) if r is not None else None,
                                                (1 if residual_normed else 0) | (2 if dual else 0), _ptr(lw) if lw is not None else None, _ptr(lb) if lb is not None else None, float(leps),
                                                _ptr(pw) if pw is not None else None, _ptr(pb) if pb is not None else None, float(peps),
                                                _ptr(pa) if pa is not None else None, parows, _ptr(y2) if y2 is not None else None,
                                                M, C, Hd, _ACTS[act], _ptr(y), _stream_ptr(x2))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "mlp_fused")
    return (y.view(x.shape), y2.view(x.shape)) if y2 is not None else y.view(x.shape)


SMALL_LINEAR_MAX_ROWS = 4096
SMALL_LINEAR_MAX_K = 256       # (a wave walks K alone, four k-steps of weights ahead: 76 us at K = 2048 against the library's 18)


def small_linear(x, weight, bias=None, rows=None, x_add=None, relu=False, residual=None, ln=None, add_features=0, transpose01=False):
    """act((x [+ x_add]) W[rows]^T + bias[rows]) [+ residual] [-> LayerNorm] for a FEW tokens in one launch (include/univs_hip.h:
    univs_small_linear_presplit_f32; csrc/small_linear.hip): the decoder's per-token Linears with `tgt + query_pos` in front and
    `norm(tgt + .)` behind (transformer_layers.py:30-46, :95-115, :150-166, :205-217).  `weight` [Nw, K] is split once and cached
    (as a whole: `rows` = (first, count) selects output features, e.g. the q / k / v thirds of `in_proj_weight`; `bias` [Nw] whole
    too); `ln` = (weight, bias, eps) needs 256 output features; `add_features`: x_add enters the first add_features outputs only (a
    multiple of 32: q, k and v of a self-attention in one launch); `transpose01` (x [A, B, K], no residual / LayerNorm): the result comes
    back as [B, A, N] contiguous (the mask embeddings [Q', T, C] -> [T, Q', C]).  Returns None when not covered (more than 4096 rows, K % 32, K > 256, N % 16,
    autograd needed): the caller keeps F.linear and the separate elementwise launches."""
    K = x.shape[-1]
    M = x.numel() // max(K, 1)
    Nw = weight.shape[0]
    f_off, N = (0, Nw) if rows is None else (int(rows[0]), int(rows[1]))
    if (not x.is_cuda or x.dtype != torch.float32 or weight.dtype != torch.float32 or weight.dim() != 2 or weight.shape[1] != K
            or M < 1 or M > SMALL_LINEAR_MAX_ROWS or K % 32 != 0 or K > SMALL_LINEAR_MAX_K or N % 16 != 0 or f_off % 4 != 0 or f_off + N > Nw
            or needs_grad(x, weight, bias, x_add, residual) or (ln is not None and N != 256) or add_features % 32 != 0
            or (transpose01 and (x.dim() != 3 or residual is not None or ln is not None))):
        return None
    x2 = x.contiguous().view(M, K)
    xa = None
    if x_add is not None:
        if x_add.dtype != torch.float32 or not x_add.is_cuda or tuple(x_add.shape) != tuple(x.shape):
            return None
        xa = x_add.contiguous()
    r = None
    if residual is not None:
        if residual.dtype != torch.float32 or not residual.is_cuda or residual.numel() != M * N or residual.shape[-1] != N:
            return None
        r = residual.contiguous()
    if bias is not None and (bias.dtype != torch.float32 or tuple(bias.shape) != (Nw,) or not bias.is_contiguous()):
        return None
    lw = lb = None
    leps = 0.0
    if ln is not None:
        lw, lb, leps = ln
        for t_ in (lw, lb):
            if t_ is not None and (t_.dtype != torch.float32 or tuple(t_.shape) != (N,) or not t_.is_cuda or not t_.is_contiguous()):
                return None
        if lw is None:
            return None
    oshape = (x.shape[1], x.shape[0], N) if transpose01 else tuple(x.shape[:-1]) + (N,)
    y = torch.empty(oshape, dtype=torch.float32, device=x.device)
    with _on(x):
        wp, winv = presplit_weights(weight)
        rc = _lib.load().univs_small_linear_presplit_f32(
            _ptr(x2), _ptr(xa) if xa is not None else None, _ptr(wp), _ptr(winv), _ptr(bias) if bias is not None else None, Nw, f_off,
            _ptr(r) if r is not None else None, _ptr(lw) if lw is not None else None, _ptr(lb) if lb is not None else None, float(leps),
            M, N, K, 1 if relu else 0, int(add_features), int(x.shape[1]) if transpose01 else 0, _ptr(y), _stream_ptr(x2))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "small_linear")
    return y


def small_mlp(x, layers_, in_ln=None, want_normed=False, transpose01=False):
    """A chain of up to three 256 -> 256 Linears (ReLU between them as flagged) on a FEW tokens in ONE launch (include/univs_hip.h:
    univs_small_mlp_presplit_f32; csrc/small_linear.hip: small_chain_kernel): the mask-embedding MLP of every prediction head
    (transformer_layers.py:205-217).  `layers_` = [(weight [256, 256], bias [256] | None, relu: bool), ...]; bit-identical to the same
    chain of `small_linear` calls.  `in_ln` = (weight, bias, eps): nn.LayerNorm(256) on the input rows inside the launch (`decoder_norm`,
    ...decoder_univs.py:513); with `want_normed` the normalised rows come back too: (y, x_normed).  `transpose01` as in `small_linear`.
    Returns None when not covered (the caller keeps the separate launches)."""
    K = x.shape[-1]
    M = x.numel() // max(K, 1)
    n = len(layers_)
    if (not x.is_cuda or x.dtype != torch.float32 or K != 256 or M < 1 or M > SMALL_LINEAR_MAX_ROWS or n < 1 or n > 3
            or (transpose01 and x.dim() != 3) or (want_normed and in_ln is None)):
        return None
    for w, b, _ in layers_:
        if (w.dtype != torch.float32 or tuple(w.shape) != (256, 256) or w._base is not None or not w.is_contiguous() or needs_grad(x, w, b)
                or (b is not None and (b.dtype != torch.float32 or tuple(b.shape) != (256,) or not b.is_contiguous()))):
            return None
    x2 = x.contiguous().view(M, K)
    lw = lb = None
    leps = 0.0
    if in_ln is not None:
        lw, lb, leps = in_ln
        for t_ in (lw, lb):
            if t_ is not None and (t_.dtype != torch.float32 or tuple(t_.shape) != (256,) or not t_.is_cuda or not t_.is_contiguous()):
                return None
        if lw is None:
            return None
    oshape = (x.shape[1], x.shape[0], 256) if transpose01 else tuple(x.shape[:-1]) + (256,)
    y = torch.empty(oshape, dtype=torch.float32, device=x.device)
    xn = torch.empty_like(x2) if want_normed else None
    P3, I3 = ctypes.c_void_p * 3, ctypes.c_int * 3
    with _on(x):
        split = [presplit_weights(w) for w, _, _ in layers_]
        wp = P3(*([_ptr(s_[0]) for s_ in split] + [None] * (3 - n)))
        wi = P3(*([_ptr(s_[1]) for s_ in split] + [None] * (3 - n)))
        bs = P3(*([(_ptr(b) if b is not None else None) for _, b, _ in layers_] + [None] * (3 - n)))
        rl = I3(*([1 if r else 0 for _, _, r in layers_] + [0] * (3 - n)))
        rc = _lib.load().univs_small_mlp_presplit_f32(_ptr(x2), n, wp, wi, bs, rl, _ptr(lw) if lw is not None else None,
                                                      _ptr(lb) if lb is not None else None, float(leps), _ptr(xn) if xn is not None else None,
                                                      M, int(x.shape[1]) if transpose01 else 0, _ptr(y), _stream_ptr(x2))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "small_mlp")
    return (y, xn.view(x.shape)) if want_normed else y


def _resident_presplit(weight, K):
    """The W-resident Linear takes the cached split image of `weight` (SWITCHES.resident_presplit; three-product arithmetic; a whole,
    contiguous tensor whose identity outlives the call -- views of parameters would be split again on every call)."""
    return (SWITCHES.resident_presplit and K % 32 == 0 and weight._base is None and weight.is_contiguous()
            and _LINEAR_TERMS != 6)


def linear_fused(x, weight, bias=None, act=None, residual=None):
    """F.linear(x, weight, bias) with a fused epilogue -- `act` in (None, 'relu', 'gelu' [the erf form; erf to 4.7e-7 absolute]) or `residual`
    (a tensor of the output's shape that is added) -- for float32 on the GPU through the three-product fp16 kernels (fp32-accurate:
    two fp16 parts per operand, three MFMA products; `configure(linear_terms=6)`: six bf16 products): the token projections of MSDeformAttn
    (ms_deform_attn.py:95-113) and of the Swin blocks (swin.py:35-58, :137-141, :163, :291-293).
    Returns None when the shape is not covered or the library GEMM is the better choice (K not a multiple of 96 / 128 or
    above 768, N % 4, fewer than 2048 rows, ranges beyond 2^31 bytes, autograd needed): the caller then keeps its own
    Linear + activation + add."""
    K = x.shape[-1]
    N = weight.shape[0]
    M = x.numel() // max(K, 1)
    if needs_grad(x, weight, bias, residual):
        return None   # autograd has to see this Linear: the library GEMM path records it
    if act not in _ACTS:
        raise RuntimeError(f"linear_fused: unknown activation {act!r}")
    if (not x.is_cuda or x.dtype != torch.float32 or weight.dtype != torch.float32 or weight.shape[1] != K
            or (K % 128 != 0 and K % 96 != 0) or K > SWITCHES.linear_kmax or N % 4 != 0 or M < 2048 or M * max(N, K) * 4 >= 2 ** 31 - 1
            or (residual is not None and _ACTS[act] != 0)):
        return None
    x2 = x.contiguous().view(M, K)
    w = weight.contiguous()
    b = bias.contiguous() if bias is not None else None
    _require_gpu("linear_fused", x2, w)
    if b is not None and (b.dtype != torch.float32 or tuple(b.shape) != (N,) or not b.is_cuda):
        raise RuntimeError("linear_fused: bias must be float32 [N] on the GPU")
    r = None
    if residual is not None:
        if residual.dtype != torch.float32 or not residual.is_cuda or tuple(residual.shape) != tuple(x.shape[:-1]) + (N,):
            raise RuntimeError("linear_fused: residual must be float32 of the output's shape on the GPU "
                               f"(got {tuple(residual.shape)}, output {tuple(x.shape[:-1]) + (N,)})")
        r = residual.contiguous()
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    with _on(x):
        rc = _lib.ERR_NOT_IMPLEMENTED
        # the streamed kernel (weights split once per tensor): wide K, and short square-ish problems (Swin stage-3 proj: 18 400
        # rows, 384 -> 384), where re-splitting the W slab in every workgroup of the resident kernel costs more than the rows
        if 0 < SWITCHES.presplit_kmin <= K or (SWITCHES.presplit_kmin > 0 and K >= 384 and N <= K and M <= 32768):
            wp, winv = presplit_weights(weight)
            rc = _lib.load().univs_linear_presplit_f32(_ptr(x2), _ptr(wp), _ptr(winv), _ptr(b) if b is not None else None,
                                                       _ptr(r) if r is not None else None, M, N, K, _ACTS[act], _ptr(y),
                                                       _stream_ptr(x2))
        if rc == _lib.ERR_NOT_IMPLEMENTED and _resident_presplit(weight, K):
            # the W-resident kernel on the split image: staging the slab is a copy (13 - 15 us per launch less than splitting it in
            # every workgroup); whole weight tensors only -- the split is cached per tensor object
            wp, winv = presplit_weights(weight)
            rc = _lib.load().univs_linear_resident_presplit_f32(_ptr(x2), _ptr(wp), _ptr(winv), _ptr(b) if b is not None else None,
                                                                _ptr(r) if r is not None else None, M, N, K, _ACTS[act], _ptr(y),
                                                                _stream_ptr(x2))
        if rc == _lib.ERR_NOT_IMPLEMENTED:
            rc = _lib.load().univs_linear_fused_f32(_ptr(x2), _ptr(w), _ptr(b) if b is not None else None,
                                                    _ptr(r) if r is not None else None, M, N, K, _ACTS[act], _ptr(y),
                                                    _stream_ptr(x2))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "linear_fused")
    return y.view(*x.shape[:-1], N)


def linear_blocked(x, weight, bias, rows_per_batch, col_block):
    """F.linear with a column-blocked output per batch element (include/univs_hip.h: univs_linear_blocked_f32):
    x [B, rows_per_batch, K] (or [B * rows_per_batch, K]) -> y [B, N / col_block, rows_per_batch, col_block].  None when the
    shape is not covered (the caller keeps the standard layout)."""
    _inference_only("linear_blocked", x, weight)
    K = x.shape[-1]
    x2 = x.reshape(-1, K).contiguous()
    w = weight.contiguous()
    Mrows, N = x2.shape[0], w.shape[0]
    if x2.dtype != torch.float32 or w.dtype != torch.float32 or w.shape[1] != K:
        return None
    _require_gpu("linear_blocked", x2, w)
    rows, cb = int(rows_per_batch), int(col_block)
    if rows < 1 or cb < 4 or cb % 4 or N % cb or Mrows % rows or K != 256:
        return None
    b = bias.contiguous() if bias is not None else None
    y = torch.empty((Mrows // rows, N // cb, rows, cb), dtype=torch.float32, device=x2.device)
    with _on(x2):
        if _resident_presplit(weight, K):
            wp, winv = presplit_weights(weight)
            rc = _lib.load().univs_linear_blocked_presplit_f32(_ptr(x2), _ptr(wp), _ptr(winv), _ptr(b) if b is not None else None, Mrows,
                                                               N, K, rows, cb, _ptr(y), _stream_ptr(x2))
        else:
            rc = _lib.load().univs_linear_blocked_f32(_ptr(x2), _ptr(w), _ptr(b) if b is not None else None, Mrows, N, K, rows, cb,
                                                      _ptr(y), _stream_ptr(x2))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "linear_blocked")
    return y


def linear_split(x, weight, bias=None, relu=False):   # (kept name: layers.linear / linear_act)
    """linear_fused with the ReLU switch of the MSDeformAttn encoder's callers."""
    return linear_fused(x, weight, bias, act="relu" if relu else None)


def mask_decode_set_impl(impl: int):
    """0 = by size (default), 1 = exact-f32 MFMA kernel, 2 = split-bf16 ("bf16 x 6") kernel where eligible."""
    _lib.check(_lib.load().univs_mask_decode_set_impl(int(impl)), "mask_decode_set_impl")


def mask_decode_last_impl() -> int:
    return int(_lib.load().univs_mask_decode_last_impl())


def mask_decode(mask_embed, mask_features):
    """einsum('tqc,tchw->qthw'): mask_embed [T,Q,C], mask_features [T,C,H,W] -> logits [Q,T,H,W]
    (== ...decoder_univs.py:527-528 for batch 1)."""
    _inference_only("mask_decode", mask_embed, mask_features)
    _require_gpu("mask_decode", mask_embed, mask_features)
    if mask_embed.dtype != torch.float32 or mask_features.dtype != torch.float32:
        raise RuntimeError("mask_decode: float32 only")
    T, Q, C = mask_embed.shape
    T2, C2, H, W = mask_features.shape
    if T2 != T or C2 != C:
        raise RuntimeError(f"mask_decode: shape mismatch {tuple(mask_embed.shape)} vs {tuple(mask_features.shape)}")
    out = torch.empty((Q, T, H, W), dtype=torch.float32, device=mask_embed.device)
    with _on(mask_embed):
        rc = _lib.load().univs_mask_decode_f32(_ptr(mask_embed), _ptr(mask_features), T, Q, C, H * W,
                                               _ptr(out), _stream_ptr(mask_embed))
    _lib.check(rc, "mask_decode")
    return out


class DeferredMask:
    """An attention mask [T, Q, hw] (uint8, 1 = key masked out) whose all-masked-row rule (...decoder_univs.py:390) has NOT been applied
    to the bytes: row r counts only where flags[r] == gen, elsewhere every key is visible (include/univs_hip.h:
    univs_mask_decode_attn_deferred_f32).  `cross_attention` consumes it as it is; `materialize()` gives the reference's bool tensor.
    Flag buffers come from a ring of four per (device, stream, row count).  Before a buffer is handed to a new mask, the mask that
    used it last -- if somebody still holds it and it was never made explicit -- is materialised (its bytes and flags are intact at
    that point of the stream), so a held DeferredMask stays valid however many masks follow it; in the decoder's order (a layer's
    cross-attention consumes its mask and drops it before the next prediction head) that never costs a launch."""
    __slots__ = ("mask", "flags", "gen", "_bool", "_entry", "__weakref__")
    dtype = torch.bool

    def __init__(self, mask, flags, gen, entry=None):
        self.mask, self.flags, self.gen, self._bool, self._entry = mask, flags, gen, None, entry

    @property
    def shape(self):
        return self.mask.shape

    @property
    def is_cuda(self):
        return True

    def dim(self):
        return self.mask.dim()

    def check_fresh(self):
        if self._bool is None and self._entry is not None and self._entry[1] != self.gen:      # (cannot happen: see mask_decode_attn)
            raise RuntimeError("DeferredMask: the row flags of this mask were overwritten by a newer generation")

    def materialize(self):
        if self._bool is None:
            self.check_fresh()
            T, Q, hw = self.mask.shape
            with _on(self.mask):
                _lib.check(_lib.load().univs_attn_mask_rows_reset(_ptr(self.mask), _ptr(self.flags), self.gen, T * Q, hw,
                                                                  _stream_ptr(self.mask)), "attn_mask_rows_reset")
            self._bool = self.mask.view(torch.bool)
        return self._bool


_MASK_FLAGS = {}          # (device, stream, rows) -> [next slot, [[flags int32 zero-initialised, last generation, weakref of its mask] x 4]]
_MASK_RING = 4


def _release_mask_flags(entry):
    """The flags buffer of `entry` is about to serve a new generation: make its last mask explicit if it is still held."""
    old = entry[2]() if entry[2] is not None else None
    if old is not None and old._bool is None:
        old.materialize()
    entry[2] = None


def mask_decode_attn(mask_embed, feat_lowres, deferred=False):
    """Fused attention-mask generation: mask_embed [T,Q,C], feat_lowres [T,C,h,w] (mask features
    resampled to the next level's size) -> bool [T,Q,h*w], True = key masked out; rows that would be
    fully masked come back all-False (...decoder_univs.py:555-566 + :390).
    `deferred=True`: a `DeferredMask` instead -- the contraction alone (no flag memset, no second pass over the mask); the rule for
    fully masked rows is applied by the consumer (`cross_attention`) from per-row generation flags."""
    _inference_only("mask_decode_attn", mask_embed, feat_lowres)
    _require_gpu("mask_decode_attn", mask_embed, feat_lowres)
    if mask_embed.dtype != torch.float32 or feat_lowres.dtype != torch.float32:
        raise RuntimeError("mask_decode_attn: float32 only")
    T, Q, C = mask_embed.shape
    T2, C2, h, w = feat_lowres.shape
    if T2 != T or C2 != C:
        raise RuntimeError("mask_decode_attn: shape mismatch")
    mask = torch.empty((T, Q, h * w), dtype=torch.uint8, device=mask_embed.device)
    if deferred and T * Q > 0 and not torch.cuda.is_current_stream_capturing():
        key = (mask_embed.device, _raw_stream(mask_embed.device.index), T * Q)
        ring = _MASK_FLAGS.get(key)
        if ring is None:
            if len(_MASK_FLAGS) > 64:
                for r_ in _MASK_FLAGS.values():                   # masks still held elsewhere keep their meaning
                    for e_ in r_[1]:
                        _release_mask_flags(e_)
                _MASK_FLAGS.clear()
            ring = _MASK_FLAGS[key] = [0, [None] * _MASK_RING]
        slot = ring[0]
        ring[0] = (slot + 1) % _MASK_RING
        e = ring[1][slot]
        if e is not None:
            _release_mask_flags(e)
        if e is None or e[1] >= 0x7FFFFFF0:
            e = ring[1][slot] = [torch.zeros((T * Q,), dtype=torch.int32, device=mask_embed.device), 0, None]
        e[1] += 1
        with _on(mask_embed):
            rc = _lib.load().univs_mask_decode_attn_deferred_f32(_ptr(mask_embed), _ptr(feat_lowres), T, Q, C, h * w, _ptr(mask),
                                                                 _ptr(e[0]), e[1], _stream_ptr(mask_embed))
        _lib.check(rc, "mask_decode_attn")
        dm = DeferredMask(mask, e[0], e[1], e)
        e[2] = weakref.ref(dm)
        return dm
    ws = torch.empty((max(T * Q, 1),), dtype=torch.int32, device=mask_embed.device)
    with _on(mask_embed):
        rc = _lib.load().univs_mask_decode_attn_f32(_ptr(mask_embed), _ptr(feat_lowres), T, Q, C, h * w,
                                                    _ptr(mask), _ptr(ws), _stream_ptr(mask_embed))
    _lib.check(rc, "mask_decode_attn")
    return mask.view(torch.bool)


def window_attention(qkv, bias, shift_mask, num_windows, scale):
    """Swin window-attention core (swin.py:137-168 between the qkv and proj linears).
    qkv [B_, Ntok, 3, nH, hd]; bias [nH, Ntok, Ntok]; shift_mask [nW, Ntok, Ntok] or None
    -> [B_, Ntok, nH*hd]."""
    _inference_only("window_attention", qkv, bias)
    _require_gpu("window_attention", qkv, bias)
    if qkv.dtype != torch.float32:
        raise RuntimeError("window_attention: float32 only")
    B_, Ntok, three, nH, hd = qkv.shape
    if three != 3 or tuple(bias.shape) != (nH, Ntok, Ntok):
        raise RuntimeError("window_attention: bad shapes")
    if shift_mask is not None:
        _require_gpu("window_attention", shift_mask)
        if tuple(shift_mask.shape) != (num_windows, Ntok, Ntok) or B_ % num_windows != 0:
            raise RuntimeError("window_attention: bad shift_mask shape")
    out = torch.empty((B_, Ntok, nH * hd), dtype=torch.float32, device=qkv.device)
    with _on(qkv):
        rc = _lib.load().univs_window_attention_f32(
            _ptr(qkv), _ptr(bias), _ptr(shift_mask) if shift_mask is not None else None, B_,
            int(num_windows), Ntok, nH, hd, float(scale), _ptr(out), _stream_ptr(qkv))
    _lib.check(rc, "window_attention")
    return out


def bilinear_pyramid3(x):
    """(bilinear_resample(x, (H/2, W/2)), (H/4, W/4), (H/8, W/8)) for float32 [..., H, W] on the GPU in ONE pass over x
    (bit-identical to the three calls): the mask features at the decoder's three attention-mask resolutions.  None when H
    or W is not a multiple of 8."""
    _inference_only("bilinear_pyramid3", x)
    x = x.contiguous()
    _require_gpu("bilinear_pyramid3", x)
    if x.dtype != torch.float32 or x.dim() < 2:
        raise RuntimeError("bilinear_pyramid3: float32 [..., H, W] only")
    H, W = x.shape[-2:]
    if H % 8 or W % 8 or H < 8 or W < 8:
        return None
    planes = x.numel() // (H * W)
    outs = [torch.empty(tuple(x.shape[:-2]) + (H >> k, W >> k), dtype=torch.float32, device=x.device) for k in (1, 2, 3)]
    with _on(x):
        rc = _lib.load().univs_bilinear_pyramid3_f32(_ptr(x), planes, H, W, _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), _stream_ptr(x))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "bilinear_pyramid3")
    return tuple(outs)


def bilinear_resample(x, size, addend=None):
    """F.interpolate(x, size=size, mode="bilinear", align_corners=False) for float32 [..., Hin, Win] on the
    GPU (decoder attention-mask path, ...decoder_univs.py:555-558); with `addend` [..., Hout, Wout] the FPN
    top-down step `addend + interpolate(x)` (msdeformattn.py:350-351) in one pass."""
    _inference_only("bilinear_resample", x, addend)
    x = x.contiguous()
    _require_gpu("bilinear_resample", x)
    if x.dtype != torch.float32 or x.dim() < 2:
        raise RuntimeError("bilinear_resample: float32 [..., H, W] only")
    Hin, Win = x.shape[-2:]
    Hout, Wout = int(size[0]), int(size[1])
    planes = x.numel() // max(Hin * Win, 1)
    oshape = tuple(x.shape[:-2]) + (Hout, Wout)
    if addend is not None:
        addend = addend.contiguous()
        _require_gpu("bilinear_resample", addend)
        if tuple(addend.shape) != oshape or addend.dtype != torch.float32:
            raise RuntimeError("bilinear_resample: addend must be float32 of the output shape")
    out = torch.empty(oshape, dtype=torch.float32, device=x.device)
    with _on(x):
        rc = _lib.load().univs_bilinear_resample_f32(_ptr(x), _ptr(addend) if addend is not None else None, _ptr(out),
                                                    planes, Hin, Win, Hout, Wout, _stream_ptr(x))
    _lib.check(rc, "bilinear_resample")
    return out


def normalize_pad(x, mean, std, size_divisibility=0, pad_to=None):
    """`F.pad((x - mean) / std, ...)` in one pass (include/univs_hip.h: univs_normalize_pad_f32): the pre-step of a clip
    (inference_video_entity.py:246-250).  x [T, C, H, W] float32 on the GPU, mean / std [C] (any broadcastable shape with C elements);
    rows / columns are zero-padded at the bottom / right up to a multiple of `size_divisibility` (or to `pad_to` = (Hp, Wp)).
    Returns None when not covered (CPU tensors, autograd, other dtypes)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4) or needs_grad(x):
        return None
    T, C, H, W = x.shape
    if pad_to is not None:
        Hp, Wp = int(pad_to[0]), int(pad_to[1])
    elif size_divisibility and size_divisibility > 1:
        d = int(size_divisibility)
        Hp, Wp = (H + d - 1) // d * d, (W + d - 1) // d * d
    else:
        Hp, Wp = H, W
    if T * C > 65535 or Hp < H or Wp < W or mean.numel() != C or std.numel() != C:
        return None
    xc = x.contiguous()
    m = mean.reshape(C).to(device=x.device, dtype=torch.float32).contiguous()
    s_ = std.reshape(C).to(device=x.device, dtype=torch.float32).contiguous()
    out = torch.empty((T, C, Hp, Wp), dtype=torch.float32, device=x.device)
    with _on(xc):
        rc = _lib.load().univs_normalize_pad_f32(_ptr(xc), _ptr(m), _ptr(s_), T, C, H, W, Hp, Wp, _ptr(out), _stream_ptr(xc))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "normalize_pad")
    return out


def group_norm_affine(x, num_groups, weight, bias, eps=1e-5):
    """GroupNorm statistics of contiguous float32 NCHW `x` on the GPU as per-plane (scale, bias) pairs [N * C, 2] with
    F.group_norm(x) == x * scale + bias (include/univs_hip.h: univs_group_norm_affine_f32) -- for `upsample2x_add`, which applies
    them while it reads x."""
    x = x.contiguous()
    _inference_only("group_norm_affine", x, weight, bias)
    _require_gpu("group_norm_affine", x, weight, bias)
    if x.dtype != torch.float32 or x.dim() < 2:
        raise RuntimeError("group_norm_affine: float32 [N, C, ...] only")
    N, C = x.shape[:2]
    if C % int(num_groups) != 0 or tuple(weight.shape) != (C,) or tuple(bias.shape) != (C,):
        raise RuntimeError("group_norm_affine: bad channel / group / parameter shapes")
    HW = x.numel() // max(N * C, 1)
    affine = torch.empty((N * C, 2), dtype=torch.float32, device=x.device)
    ws = torch.empty(N * C * 2 * max(1, (HW + 8191) // 8192), dtype=torch.float32, device=x.device)
    with _on(x):
        rc = _lib.load().univs_group_norm_affine_f32(_ptr(x), _ptr(weight.contiguous()), _ptr(bias.contiguous()), N, C, HW,
                                                    int(num_groups), float(eps), _ptr(ws), ws.numel(), _ptr(affine), _stream_ptr(x))
    _lib.check(rc, "group_norm_affine")
    return affine


def upsample2x_add(x, addend, affine=None):
    """addend + F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False) for float32 [..., H, W] / [..., 2 H, 2 W] on
    the GPU: the FPN top-down step (msdeformattn.py:350-351), bit-identical to `bilinear_resample(x, size, addend)`; with `affine`
    [planes, 2] (`group_norm_affine` of the addend) the addend is normalised on the way in.  None when the shape is not covered
    (odd W, a size that is not exactly twice the input's)."""
    if (not x.is_cuda or x.dtype != torch.float32 or addend.dtype != torch.float32 or x.dim() < 2 or needs_grad(x, addend)
            or tuple(addend.shape[:-2]) != tuple(x.shape[:-2]) or addend.shape[-2] != 2 * x.shape[-2]
            or addend.shape[-1] != 2 * x.shape[-1] or x.shape[-1] % 2 != 0):
        return None
    x, addend = x.contiguous(), addend.contiguous()
    Hin, Win = x.shape[-2:]
    planes = x.numel() // max(Hin * Win, 1)
    if affine is not None and (affine.dtype != torch.float32 or tuple(affine.shape) != (planes, 2) or not affine.is_contiguous()):
        raise RuntimeError("upsample2x_add: affine must be contiguous float32 [planes, 2]")
    out = torch.empty_like(addend)
    with _on(x):
        rc = _lib.load().univs_upsample2x_add_f32(_ptr(x), _ptr(addend), _ptr(affine) if affine is not None else None, _ptr(out),
                                                 planes, Hin, Win, _stream_ptr(x))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "upsample2x_add")
    return out


def conv3x3(x, weight):
    """F.conv2d(x, weight, None, stride=1, padding=1) for a 3 x 3 kernel, float32 NCHW on the GPU, through the three-product fp16
    streamed GEMM with tap addressing on weights split once per tensor (the FPN output convolution, msdeformattn.py:227-232).  Returns None when the
    shape is not covered: the caller keeps the library convolution."""
    if (not x.is_cuda or x.dtype != torch.float32 or weight.dtype != torch.float32 or x.dim() != 4
            or tuple(weight.shape[2:]) != (3, 3) or weight.shape[1] != x.shape[1] or needs_grad(x, weight)):
        return None
    T, Cin, H, W = x.shape
    Cout = weight.shape[0]
    x = x.contiguous()
    y = torch.empty((T, Cout, H, W), dtype=torch.float32, device=x.device)
    with _on(x):
        rc = _lib.ERR_NOT_IMPLEMENTED
        if SWITCHES.presplit_kmin > 0:
            wp, winv = presplit_weights(weight, conv=True)
            rc = _lib.load().univs_conv3x3_presplit_f32(_ptr(x), _ptr(wp), _ptr(winv), T, Cin, Cout, H, W, _ptr(y), _stream_ptr(x))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "conv3x3")
    return y


def conv3x3_nhwc(x_nhwc, weight):
    """`conv3x3` on a channels-last operand x [T, H, W, Cin] (contiguous) -> NCHW [T, Cout, H, W], bit-identical to
    conv3x3(x.permute(0, 3, 1, 2)) (include/univs_hip.h: univs_conv3x3_nhwc_presplit_f32).  None when not covered."""
    if (not x_nhwc.is_cuda or x_nhwc.dtype != torch.float32 or weight.dtype != torch.float32 or x_nhwc.dim() != 4 or not x_nhwc.is_contiguous()
            or tuple(weight.shape[2:]) != (3, 3) or weight.shape[1] != x_nhwc.shape[3] or needs_grad(x_nhwc, weight) or SWITCHES.presplit_kmin <= 0):
        return None
    T, H, W, Cin = x_nhwc.shape
    Cout = weight.shape[0]
    y = torch.empty((T, Cout, H, W), dtype=torch.float32, device=x_nhwc.device)
    with _on(x_nhwc):
        wp, winv = presplit_weights(weight, conv=True)
        rc = _lib.load().univs_conv3x3_nhwc_presplit_f32(_ptr(x_nhwc), _ptr(wp), _ptr(winv), T, Cin, Cout, H, W, _ptr(y), _stream_ptr(x_nhwc))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "conv3x3_nhwc")
    return y


def conv1x1(x, weight, bias=None):
    """F.conv2d(x, weight, bias) for a 1 x 1 kernel (stride 1, no padding), float32 NCHW on the GPU, through the three-product fp16
    streamed GEMM (include/univs_hip.h: univs_conv1x1_presplit_f32) with the bias in the epilogue: the lateral, mask-feature and
    input-projection convolutions of the pixel decoder (msdeformattn.py:205-232, :262-283).  None when not covered."""
    if (not x.is_cuda or x.dtype != torch.float32 or weight.dtype != torch.float32 or x.dim() != 4 or weight.dim() != 4
            or tuple(weight.shape[2:]) != (1, 1) or weight.shape[1] != x.shape[1] or needs_grad(x, weight, bias)
            or SWITCHES.presplit_kmin <= 0):
        return None
    T, Cin, H, W = x.shape
    Cout = weight.shape[0]
    if (Cin % 96 and Cin % 128) or Cout % 16 or T * H * W < 4096:
        return None
    if bias is not None and (bias.dtype != torch.float32 or tuple(bias.shape) != (Cout,) or not bias.is_cuda):
        return None
    x = x.contiguous()
    b = bias.contiguous() if bias is not None else None
    y = torch.empty((T, Cout, H, W), dtype=torch.float32, device=x.device)
    with _on(x):
        wp, winv = presplit_weights(weight)
        rc = _lib.load().univs_conv1x1_presplit_f32(_ptr(x), _ptr(wp), _ptr(winv), _ptr(b) if b is not None else None, T, Cin, Cout, H, W,
                                                    _ptr(y), _stream_ptr(x))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "conv1x1")
    return y


def patch_embed4(x, weight, bias=None, ln=None):
    """Swin PatchEmbed in one pass (include/univs_hip.h: univs_patch_embed4_f32): the 4 x 4 / stride-4 convolution of a 3-channel
    image + bias, tokens in [T, H/4 * W/4, E] order, optionally LayerNorm `ln` = (weight, bias, eps) on each token
    (swin.py:307-339).  x [T, 3, H, W] with H, W multiples of 4.  None when not covered."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4) or needs_grad(x, weight, bias):
        return None
    T, Cin, H, W = x.shape
    E = weight.shape[0]
    if Cin != 3 or tuple(weight.shape[1:]) != (3, 4, 4) or H % 4 or W % 4 or E not in (96, 128, 192):
        return None
    lw = lb = None
    leps = 0.0
    if ln is not None:
        lw, lb, leps = ln
    x, weight = x.contiguous(), weight.contiguous()
    out = torch.empty((T, (H // 4) * (W // 4), E), dtype=torch.float32, device=x.device)
    with _on(x):
        rc = _lib.load().univs_patch_embed4_f32(_ptr(x), _ptr(weight), _ptr(bias) if bias is not None else None,
                                                _ptr(lw) if lw is not None else None, _ptr(lb) if lb is not None else None, float(leps),
                                                T, H, W, E, _ptr(out), _stream_ptr(x))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "patch_embed4")
    return out


def decoder_memory(x, level_embed, pos_yx, pos_t):
    """(memory, key) [HW, T, C] of one feature level for the decoder's cross-attention, from the NCHW features in one pass
    (include/univs_hip.h: univs_decoder_memory_f32): memory = x transposed + level_embed, key = memory + (pos_yx + pos_t).
    x [T, C, H, W] (or [T, C, HW]); level_embed [C]; pos_yx [HW, C]; pos_t [T, C].  None when not covered."""
    if not (x.is_cuda and x.dtype == torch.float32) or needs_grad(x, level_embed):
        return None
    T, C = x.shape[:2]
    HW = x.numel() // max(T * C, 1)
    if C % 4 or HW % 4 or tuple(level_embed.shape) != (C,) or tuple(pos_yx.shape) != (HW, C) or tuple(pos_t.shape) != (T, C):
        return None
    x, level_embed, pos_yx, pos_t = x.contiguous(), level_embed.contiguous(), pos_yx.contiguous(), pos_t.contiguous()
    mem = torch.empty((HW, T, C), dtype=torch.float32, device=x.device)
    key = torch.empty((HW, T, C), dtype=torch.float32, device=x.device)
    with _on(x):
        rc = _lib.load().univs_decoder_memory_f32(_ptr(x), _ptr(level_embed), _ptr(pos_yx), _ptr(pos_t), T, C, HW, _ptr(mem), _ptr(key),
                                                  _stream_ptr(x))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "decoder_memory")
    return mem, key


def transpose_last2(x):
    """Contiguous copy of `x.transpose(-2, -1)` for a float32 tensor on the GPU (LDS tile transpose at HBM rate instead of
    ATen's strided copy): tokens [B, H*W, C] <-> channel-major [B, C, H*W] at the edges of the Swin backbone
    (swin.py:331-336, :676-683).  Any leading dimensions; a 3-D input may be a row range x[:, r0:r1, :] of a wider contiguous
    tensor (the per-level split of the encoder output): only its batch stride is then not dense, and no copy is made first.
    Falls back to ATen for shapes the kernel does not cover."""
    if (not x.is_cuda or x.dtype != torch.float32 or x.dim() < 2 or (torch.is_grad_enabled() and x.requires_grad)):
        return x.transpose(-2, -1).contiguous()
    R, C = x.shape[-2], x.shape[-1]
    bstride = 0
    if not x.is_contiguous():
        if (x.dim() == 3 and x.stride(2) == 1 and x.stride(1) == C and x.stride(0) >= R * C and x.stride(0) % 4 == 0
                and x.data_ptr() % 16 == 0):
            bstride = x.stride(0)
        else:
            x = x.contiguous()
    B = x.numel() // max(R * C, 1)
    out = torch.empty(x.shape[:-2] + (C, R), dtype=torch.float32, device=x.device)
    with _on(x):
        rc = _lib.load().univs_transpose_strided_f32(_ptr(x), B, R, C, bstride, _ptr(out), _stream_ptr(x))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return x.transpose(-2, -1).contiguous()
    _lib.check(rc, "transpose_last2")
    return out


def tokens_from_nchw(xs, affines, lvl_pos):
    """The encoder input of the pixel decoder from the levels' NCHW maps in one launch per level (include/univs_hip.h:
    univs_transpose_ex_f32): xs[l] [T, C, H_l, W_l] float32 on the GPU, affines[l] [T * C, 2] or None (`group_norm_affine` of xs[l]: the
    GroupNorm of `input_proj` applied on the way through), lvl_pos [1, S, C] or None -> (src_flatten [T, S, C], src_flatten + lvl_pos or
    None): `torch.cat([x.flatten(2).transpose(1, 2) ...], 1)` and the first layer's `with_pos_embed` (msdeformattn.py:168-188, :61-63)
    without the concatenation and the add as passes of their own.  None when a level is not covered (HW or C not a multiple of 4)."""
    T, C = xs[0].shape[:2]
    hws = [int(x.shape[2]) * int(x.shape[3]) for x in xs]
    S = sum(hws)
    if any((not x.is_cuda) or x.dtype != torch.float32 or x.dim() != 4 or tuple(x.shape[:2]) != (T, C) or hw % 4 != 0
           for x, hw in zip(xs, hws)) or C % 4 != 0 or T > 65535 or needs_grad(*xs):
        return None
    if lvl_pos is not None and (tuple(lvl_pos.shape[-2:]) != (S, C) or lvl_pos.numel() != S * C or lvl_pos.dtype != torch.float32
                                or not lvl_pos.is_contiguous()):
        raise RuntimeError("tokens_from_nchw: lvl_pos must be contiguous float32 [1, S, C]")
    src = torch.empty((T, S, C), dtype=torch.float32, device=xs[0].device)
    q0 = torch.empty_like(src) if lvl_pos is not None else None
    lib = _lib.load()
    r0 = 0
    with _on(src):
        for x, aff, hw in zip(xs, affines, hws):
            x = x.contiguous()
            if aff is not None and (aff.dtype != torch.float32 or tuple(aff.shape) != (T * C, 2) or not aff.is_contiguous()):
                raise RuntimeError("tokens_from_nchw: affine must be contiguous float32 [T * C, 2]")
            off = r0 * C * 4
            rc = lib.univs_transpose_ex_f32(_ptr(x), T, C, hw, 0, _ptr(aff) if aff is not None else None, _ptr(src) + off, S * C,
                                            (_ptr(lvl_pos) + off) if lvl_pos is not None else None,
                                            (_ptr(q0) + off) if q0 is not None else None, _stream_ptr(src))
            if rc == _lib.ERR_NOT_IMPLEMENTED:
                return None
            _lib.check(rc, "tokens_from_nchw")
            r0 += hw
    return src, q0


def patch_merge_norm(x, weight, bias, eps=1e-5):
    """Swin PatchMerging up to its Linear in one pass (include/univs_hip.h: univs_patch_merge_norm_f32): x [B, H, W, C] float32 on the GPU
    -> LayerNorm over the 4 C channels of the 2 x 2 patches [B, ceil(H/2) * ceil(W/2), 4 C], channel order and zero padding of odd sizes as
    PatchMerging.forward (swin.py:341-386).  None when the width is not covered (C % 4, C > 768) or autograd is needed."""
    if (not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or needs_grad(x, weight, bias) or x.shape[-1] % 4 != 0
            or x.shape[-1] > 768):
        return None
    B, H, W, C = x.shape
    x, weight, bias = x.contiguous(), weight.contiguous(), bias.contiguous()
    if tuple(weight.shape) != (4 * C,) or tuple(bias.shape) != (4 * C,):
        raise RuntimeError("patch_merge_norm: weight / bias must be [4 C]")
    out = torch.empty((B, ((H + 1) // 2) * ((W + 1) // 2), 4 * C), dtype=torch.float32, device=x.device)
    with _on(x):
        rc = _lib.load().univs_patch_merge_norm_f32(_ptr(x), _ptr(weight), _ptr(bias), B, H, W, C, float(eps), _ptr(out), _stream_ptr(x))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "patch_merge_norm")
    return out


def layer_norm(x, weight, bias, eps=1e-5, residual=None, return_sum=False, post_add=None):
    """LayerNorm over the last dimension of contiguous float32 `x` on the GPU, optionally of
    `x + residual` (and then optionally also returning that sum): nn.LayerNorm in the Swin blocks
    (swin.py:236-262), encoder layers (msdeformattn.py:61-95) and decoder layers.
    `post_add` (with a residual, without return_sum): additionally returns `out + post_add` -- the encoder's
    `with_pos_embed(src, pos)` for the next layer, from the same pass.
    Returns `out`, `(x + residual, out)` or `(out, out + post_add)`."""
    x, weight, bias = x.contiguous(), weight.contiguous(), bias.contiguous()   # views (e.g. NCHW -> tokens) are copied once
    _inference_only("layer_norm", x, weight, bias, residual)
    _require_gpu("layer_norm", x, weight, bias)
    if x.dtype != torch.float32:
        raise RuntimeError("layer_norm: float32 only")
    if return_sum and residual is None:
        raise RuntimeError("layer_norm: return_sum needs a residual")
    C = x.shape[-1]
    if tuple(weight.shape) != (C,) or tuple(bias.shape) != (C,):
        raise RuntimeError("layer_norm: weight / bias must be [C]")
    if residual is not None:
        residual = residual.contiguous()
        _require_gpu("layer_norm", residual)
        if residual.shape != x.shape or residual.dtype != torch.float32:
            raise RuntimeError("layer_norm: residual must match x")
    out = torch.empty_like(x)
    s = torch.empty_like(x) if return_sum else None
    rows = x.numel() // max(C, 1)
    if post_add is not None:
        if residual is None or return_sum:
            raise RuntimeError("layer_norm: post_add needs a residual and excludes return_sum")
        post_add = post_add.contiguous()
        _require_gpu("layer_norm", post_add)
        # same shape, or broadcast over the leading dimensions ([1, S, C] position embeddings against [N, S, C] tokens)
        arows = post_add.numel() // max(C, 1)
        lead = x.dim() - post_add.dim()
        ok = (post_add.dtype == torch.float32 and lead >= 0 and post_add.shape[-1] == C and rows % max(arows, 1) == 0
              and all(a == b or (i < post_add.dim() - 1 and all(int(d) == 1 for d in post_add.shape[:i + 1]))
                      for i, (a, b) in enumerate(zip(post_add.shape, x.shape[lead:]))))
        if not ok:
            raise RuntimeError("layer_norm: post_add must match x or broadcast over its leading dimensions")
        out2 = torch.empty_like(x)
        with _on(x):
            rc = _lib.load().univs_layer_norm_add_f32(_ptr(x), _ptr(residual), _ptr(weight), _ptr(bias), _ptr(post_add), arows, rows,
                                                      C, float(eps), None, _ptr(out), _ptr(out2), _stream_ptr(x))
        _lib.check(rc, "layer_norm")
        return out, out2
    with _on(x):
        rc = _lib.load().univs_layer_norm_f32(_ptr(x), _ptr(residual) if residual is not None else None,
                                             _ptr(weight.contiguous()), _ptr(bias.contiguous()), rows, C, float(eps),
                                             _ptr(s) if s is not None else None, _ptr(out), _stream_ptr(x))
    _lib.check(rc, "layer_norm")
    return (s, out) if return_sum else out


def group_norm(x, num_groups, weight, bias, eps=1e-5, relu=False):
    """F.group_norm(x, num_groups, weight, bias, eps) [+ relu] for contiguous float32 NCHW `x` on the GPU: the
    Conv2d(norm=GN, activation=relu) epilogues of the pixel decoder (msdeformattn.py:214-232, :262-283)."""
    x = x.contiguous()
    _inference_only("group_norm", x, weight, bias)
    _require_gpu("group_norm", x, weight, bias)
    if x.dtype != torch.float32 or x.dim() < 2:
        raise RuntimeError("group_norm: float32 [N, C, ...] only")
    N, C = x.shape[:2]
    if C % int(num_groups) != 0 or tuple(weight.shape) != (C,) or tuple(bias.shape) != (C,):
        raise RuntimeError("group_norm: bad channel / group / parameter shapes")
    HW = x.numel() // max(N * C, 1)
    out = torch.empty_like(x)
    ws = torch.empty(N * C * 2 * max(1, (HW + 8191) // 8192), dtype=torch.float32, device=x.device)
    with _on(x):
        rc = _lib.load().univs_group_norm_f32(_ptr(x), _ptr(weight.contiguous()), _ptr(bias.contiguous()), N, C, HW,
                                             int(num_groups), float(eps), 1 if relu else 0, _ptr(ws), ws.numel(),
                                             _ptr(out), _stream_ptr(x))
    _lib.check(rc, "group_norm")
    return out


def masked_softmax_(scores, mask=None):
    """In-place softmax over the last dimension of contiguous float32 attention scores [N, h, L, S] with an
    optional boolean / uint8 mask [N, L, S] (True = masked out, shared by the heads): the masked softmax inside
    nn.MultiheadAttention (transformer_layers.py:101-105).  Returns `scores`."""
    _inference_only("masked_softmax_", scores)
    _require_gpu("masked_softmax_", scores)
    if scores.dtype != torch.float32 or scores.dim() != 4:
        raise RuntimeError("masked_softmax_: float32 [N, h, L, S] only")
    N, h, L, S = scores.shape
    mptr = None
    if mask is not None:
        _require_gpu("masked_softmax_", mask)
        if tuple(mask.shape) != (N, L, S) or mask.dtype not in (torch.bool, torch.uint8):
            raise RuntimeError("masked_softmax_: mask must be bool / uint8 [N, L, S]")
        mptr = _ptr(mask)
    with _on(scores):
        rc = _lib.load().univs_masked_softmax_f32(_ptr(scores), mptr, N, h, L, S, _stream_ptr(scores))
    _lib.check(rc, "masked_softmax_")
    return scores


def proca_attention(qkv0, kd, vd, num_heads):
    """ProCA attention (include/univs_hip.h: univs_proca_attention_f32; csrc/proca_attn.hip): qkv0 [Q_p * T, 3 E] (query, first key,
    first value of every (prompt query, frame)), kd / vd [Q_p, L, T, E] (the dense prompt tokens' key / value projections) ->
    [Q_p * T, E] = softmax(q [k0; kd]^T / sqrt(d)) [v0; vd].  None when the shape is not covered (head_dim != 32)."""
    _inference_only("proca_attention", qkv0, kd, vd)
    _require_gpu("proca_attention", qkv0, kd, vd)
    Qp, L, T, E = kd.shape
    h = int(num_heads)
    if (any(t.dtype != torch.float32 or not t.is_contiguous() for t in (qkv0, kd, vd)) or tuple(vd.shape) != tuple(kd.shape)
            or tuple(qkv0.shape) != (Qp * T, 3 * E) or E % h):
        raise RuntimeError("proca_attention: contiguous float32 qkv0 [Q_p T, 3 E], kd / vd [Q_p, L, T, E]")
    out = torch.empty((Qp * T, E), dtype=torch.float32, device=kd.device)
    with _on(kd):
        rc = _lib.load().univs_proca_attention_f32(_ptr(qkv0), _ptr(kd), _ptr(vd), Qp, L, T, h, E // h, 1.0 / math.sqrt(E // h), _ptr(out),
                                                   _stream_ptr(kd))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "proca_attention")
    return out


def prompt_prefix(masks, boxes, scale, mask_thresh=0.5):
    """The annotation-only part of VisualPromptEncoder.get_mask_prompt for F key frames x n entities in three launches
    (include/univs_hip.h: univs_prompt_prefix_f32; csrc/prompt_sampler.hip).  masks [F, n, h, w] float32, boxes [F, n, 4] normalised
    xyxy -> the dict of `annotation_prefix` (univs_amd/modeling/prompt_encoder.py), bit for bit."""
    _inference_only("prompt_prefix", masks, boxes)
    masks, boxes = masks.contiguous(), boxes.contiguous()
    _require_gpu("prompt_prefix", masks, boxes)
    Fk, n, h, w = masks.shape
    if masks.dtype != torch.float32 or boxes.dtype != torch.float32 or tuple(boxes.shape) != (Fk, n, 4) or h % scale or w % scale:
        raise RuntimeError("prompt_prefix: float32 masks [F, n, h, w] with h, w multiples of the scale, boxes [F, n, 4]")
    dev, N, hi, wi = masks.device, Fk * n, h // scale, w // scale
    feat_masks = torch.empty((Fk, n, hi, wi), dtype=torch.float32, device=dev)
    stats = torch.zeros(2 * N + Fk, dtype=torch.int32, device=dev)
    sel = torch.empty((Fk, n, h, w), dtype=torch.bool, device=dev)
    rowcnt = torch.empty((Fk, n, h), dtype=torch.int32, device=dev)
    fmb = torch.empty((Fk, n, hi, wi), dtype=torch.bool, device=dev)
    counts = torch.empty((Fk, 2 * n), dtype=torch.int32, device=dev)
    flags = torch.empty((2, Fk, n), dtype=torch.bool, device=dev)
    with _on(masks):
        rc = _lib.load().univs_prompt_prefix_f32(_ptr(masks), _ptr(boxes), Fk, n, h, w, int(scale), float(mask_thresh), _ptr(feat_masks),
                                                 _ptr(stats), _ptr(sel), _ptr(rowcnt), _ptr(fmb), _ptr(counts), _ptr(flags[0]),
                                                 _ptr(flags[1]), _stream_ptr(masks))
    _lib.check(rc, "prompt_prefix")
    return {"valid": flags[0], "visible": flags[1], "feat_masks": feat_masks, "feat_masks_binary": fmb, "sel": sel, "rowcnt": rowcnt,
            "counts": counts}


def prompt_draw(pre, R, u=None, keys=None, tab=None):
    """The draws of a clip's key frames -> pixels, one launch (univs_prompt_draw).  `pre` = the dict of `prompt_prefix` /
    `annotation_prefix`; either u [N, 1] and keys [N, HW] (uniform numbers of the device generator) or tab [N, R + 2] int64 (the
    reference's host draws: R dense ranks, the "empty" flag, the point's rank).  Returns (point_idx [N] int64, point_coords [N, 2],
    dense_idx [N, R] int64, empty [N] bool), or None when the shape is not covered."""
    sel, rowcnt, fmb, counts = pre["sel"], pre["rowcnt"], pre["feat_masks_binary"], pre["counts"]
    _require_gpu("prompt_draw", sel)
    Fk, n, h, w = sel.shape
    N, HW = Fk * n, fmb.shape[-2] * fmb.shape[-1]
    if any(not t.is_contiguous() for t in (sel, rowcnt, fmb, counts)) or sel.dtype != torch.bool or fmb.dtype != torch.bool \
            or rowcnt.dtype != torch.int32 or counts.dtype != torch.int32:
        return None
    if tab is not None:
        if tab.dtype != torch.int64 or tuple(tab.shape) != (N, R + 2):
            raise RuntimeError("prompt_draw: tab [N, R + 2] int64")
        tab = tab.contiguous()
    else:
        if u.dtype != torch.float32 or keys.dtype != torch.float32 or u.numel() != N or tuple(keys.shape) != (N, HW) or HW < R:
            return None
        u, keys = u.contiguous(), keys.contiguous()
    dev = sel.device
    point_idx = torch.empty(N, dtype=torch.int64, device=dev)
    dense_idx = torch.empty((N, R), dtype=torch.int64, device=dev)
    empty = torch.empty(N, dtype=torch.bool, device=dev)
    coords = torch.empty((N, 2), dtype=torch.float32, device=dev)
    null = ctypes.c_void_p(0)
    with _on(sel):
        rc = _lib.load().univs_prompt_draw(_ptr(sel), _ptr(rowcnt), _ptr(fmb), _ptr(counts), null if tab is not None else _ptr(u),
                                           null if tab is not None else _ptr(keys), null if tab is None else _ptr(tab), Fk, n, h, w, HW,
                                           int(R), _ptr(point_idx), _ptr(dense_idx), _ptr(empty), _ptr(coords), _stream_ptr(sel))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "prompt_draw")
    return point_idx, coords, dense_idx, empty


def prompt_point_pe(xy, z, dim_t, dim_tz, scale, n):
    """Position tokens of F n sampled points in one launch (univs_prompt_point_pe_f32): xy [F n, 2], z [F] (scaled frame coordinate),
    the frequency vectors dim_t [Fq] / dim_tz [2 Fq] -> [F n, 2 Fq], the bits of position_encoding._points."""
    N, Fk, Fq = xy.shape[0], z.shape[0], dim_t.shape[0]
    ts = (xy, z, dim_t, dim_tz)
    if any(t.dtype != torch.float32 or not t.is_cuda for t in ts) or N != Fk * n or dim_tz.shape[0] != 2 * Fq or tuple(xy.shape) != (N, 2):
        return None
    xy, z, dim_t, dim_tz = (t.contiguous() for t in ts)
    out = torch.empty((N, 2 * Fq), dtype=torch.float32, device=xy.device)
    with _on(xy):
        rc = _lib.load().univs_prompt_point_pe_f32(_ptr(xy), _ptr(z), _ptr(dim_t), _ptr(dim_tz), float(scale), Fk, int(n), Fq, _ptr(out),
                                                   _stream_ptr(xy))
    _lib.check(rc, "prompt_point_pe")
    return out


def mask_stats(x, t_hi=1.0, t_lo=-1.0, t_box=0.0, valid=None):
    """Per-plane statistics of mask logits in one pass (include/univs_hip.h: univs_mask_stats_f32 / _strided_f32; csrc/mask_stats.hip): x
    [..., H, W] float32 on the GPU -> int32 [..., 8] = (|{x > t_hi}|, |{x > t_lo}|, left, top, right, bottom of {x > t_box} -- inclusive,
    zeros when empty --, non-empty, 0) over rows [0, valid[0]) x columns [0, valid[1]) (the whole plane by default): what
    `calculate_mask_quality_scores` and `convert_mask_to_box` (utils/comm.py) compute with ~25 launches.  x is contiguous, or a 4-D view
    [N, T, H, W] whose planes are dense and whose two leading strides are free (`history[:, -T:]`: no copy).  None when not covered
    (more than 65 535 planes, autograd needed)."""
    if not x.is_cuda:
        raise RuntimeError(f"mask_stats: Not implemented on the CPU (tensor on {x.device}); the HIP extension is the only implementation")
    if x.dtype != torch.float32 or x.dim() < 2:
        raise RuntimeError("mask_stats: float32 [..., H, W] only")
    H, W = int(x.shape[-2]), int(x.shape[-1])
    hv, wv = (H, W) if valid is None else (min(int(valid[0]), H), min(int(valid[1]), W))
    planes = x.numel() // max(H * W, 1)
    if planes > 65535 or needs_grad(x) or H * W == 0:
        return None
    dense_planes = x.stride(-1) == 1 and x.stride(-2) == W
    if x.is_contiguous():
        outer, inner, so, si = (1 if planes else 0), max(planes, 1), 0, H * W
    elif x.dim() == 4 and dense_planes and x.stride(1) >= H * W and x.stride(0) >= 0:
        outer, inner, so, si = int(x.shape[0]), int(x.shape[1]), int(x.stride(0)), int(x.stride(1))
    else:
        raise RuntimeError("mask_stats: all tensors have to be contiguous (or a [N, T, H, W] view with dense planes)")
    out = torch.empty(tuple(x.shape[:-2]) + (8,), dtype=torch.int32, device=x.device)
    if planes == 0:
        return out
    with _on(x):
        rc = _lib.load().univs_mask_stats_strided_f32(_ptr(x), outer, inner, so, si, H, W, hv, wv, float(t_hi), float(t_lo), float(t_box),
                                                      _ptr(out), _stream_ptr(x))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "mask_stats")
    return out


def token_mean(x, add=None):
    """Mean over the non-blank tokens (univs_token_mean_f32): x [n, L, T, C] -> [n, T, C] = x.sum(1) / max(1, number of tokens l whose
    C channels are not all zero) (+ add [C]).  None when not covered."""
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or needs_grad(x, add):
        return None
    n, L, T, C = x.shape
    if add is not None and (add.dtype != torch.float32 or add.numel() != C or not add.is_cuda):
        return None
    x = x.contiguous()
    a = add.contiguous().view(-1) if add is not None else None
    out = torch.empty((n, T, C), dtype=torch.float32, device=x.device)
    with _on(x):
        rc = _lib.load().univs_token_mean_f32(_ptr(x), _ptr(a) if a is not None else None, n, L, T, C, _ptr(out), _stream_ptr(x))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "token_mean")
    return out


def _fcp_strides(t):
    """(frame, channel, pixel) element strides of a [F, C, h, w] map whose pixels are evenly spaced (dense or channels-last)"""
    if t.stride(2) != t.shape[3] * t.stride(3):
        return None
    return (t.stride(0), t.stride(1), t.stride(3))


def prompt_tokens(img_features, img_pos, query_feats, query_pe, dense_idx, empty, valid, boxes, kf, T):
    """The dense prompt tokens and cross-attention masks of a clip's key frames in two launches (univs_prompt_tokens_f32):
    img_features / img_pos [F, C, h_img, w_img], query_feats / query_pe [N, C] (the pooled tokens, used for empty masks), dense_idx
    [N, R], empty / valid [N] bool, boxes [F, n, 4], kf [F] int64 (the key frame's position in the clip) ->
    fd, pd [F, n, R, T, C], attn [F, T, 1, n, HW] bool.  None when a layout is not covered."""
    _inference_only("prompt_tokens", img_features, img_pos)
    if not (img_features.is_cuda and img_pos.is_cuda):
        raise RuntimeError("prompt_tokens: Not implemented on the CPU; the HIP extension is the only implementation")
    Fk, C, hi, wi = img_features.shape
    n = boxes.shape[1]
    N, R = dense_idx.shape
    fs, ps = _fcp_strides(img_features), _fcp_strides(img_pos)
    if (fs is None or ps is None or img_features.dtype != torch.float32 or img_pos.dtype != torch.float32 or tuple(img_pos.shape) != (Fk, C, hi, wi)
            or N != Fk * n or tuple(query_feats.shape) != (N, C) or tuple(query_pe.shape) != (N, C) or query_feats.dtype != torch.float32
            or query_pe.dtype != torch.float32 or empty.dtype != torch.bool or valid.dtype != torch.bool or kf.dtype != torch.int64):
        return None
    query_feats, query_pe, dense_idx, boxes = query_feats.contiguous(), query_pe.contiguous(), dense_idx.contiguous(), boxes.contiguous()
    empty, valid, kf = empty.contiguous().view(-1), valid.contiguous().view(-1), kf.contiguous()
    dev = img_features.device
    fd = torch.empty((Fk, n, R, T, C), dtype=torch.float32, device=dev)
    pd = torch.empty((Fk, n, R, T, C), dtype=torch.float32, device=dev)
    attn = torch.empty((Fk, T, 1, n, hi * wi), dtype=torch.bool, device=dev)
    fs_c, ps_c = (ctypes.c_int64 * 3)(*fs), (ctypes.c_int64 * 3)(*ps)
    with _on(img_features):
        rc = _lib.load().univs_prompt_tokens_f32(_ptr(img_features), ctypes.cast(fs_c, ctypes.c_void_p), _ptr(img_pos),
                                                 ctypes.cast(ps_c, ctypes.c_void_p), _ptr(query_feats), _ptr(query_pe), _ptr(dense_idx),
                                                 _ptr(empty), _ptr(valid), _ptr(boxes), _ptr(kf), Fk, n, R, int(T), C, hi, wi, _ptr(fd),
                                                 _ptr(pd), _ptr(attn), _stream_ptr(img_features))
    _lib.check(rc, "prompt_tokens")
    return fd, pd, attn


def _seq_first_ld(t, N, E):
    """Leading dimension (floats between batch entries) of a sequence-first [S, N, E] tensor that is dense or a column slice of a
    wider dense [S, N, E'] tensor; None when the layout is anything else."""
    if t.stride(-1) != 1 or t.dim() != 3:
        return None
    ld = t.stride(1) if N > 1 else (t.stride(0) if t.shape[0] > 1 else E)
    if ld < E or ld % 4 or (t.shape[0] > 1 and t.stride(0) != N * ld) or (t.data_ptr() & 15):
        return None
    return ld


_PAD4 = {}      # id(mask) -> (weak reference to the mask, its version, its rows padded to a multiple of four bytes)


def pad4_mask(mask):
    """[..., S] bool / uint8 -> [..., (S + 3) & ~3], zero-padded: the attention kernel reads mask rows in aligned dwords.  Cached per
    tensor OBJECT and version -- the decoder's self-attention mask is one cached tensor for all layers and clips of a shape -- so
    callers pass the tensor they hold, not a view made per call."""
    if mask.shape[-1] % 4 == 0:
        return mask.contiguous()
    hit = _PAD4.get(id(mask))
    if hit is not None and hit[0]() is mask and hit[1] == mask._version:
        return hit[2]
    padded = torch.nn.functional.pad(mask, (0, (-mask.shape[-1]) % 4)).contiguous()
    if len(_PAD4) > 32:
        _PAD4.clear()
    _PAD4[id(mask)] = (weakref.ref(mask), mask._version, padded)
    return padded


def cross_attention(q, k, v, mask, num_heads, scale):
    """softmax(scale q k^T, masked) v per (batch entry, head) in one pass over the keys (include/univs_hip.h:
    univs_cross_attention_f32; csrc/cross_attn.hip): the attention core of nn.MultiheadAttention as the decoder's
    CrossAttentionLayer uses it (transformer_layers.py:95-115) between the in- and out-projections.
    q [L, N, E], k / v [S, N, E] sequence-first float32 (E = num_heads * 32), dense or column slices of wider projections; mask
    bool / uint8 [N, L, S] (True = masked out, shared by the heads), a `DeferredMask` of that shape, or None.
    Returns [L, N, E], or None when not covered."""
    _inference_only("cross_attention", q, k, v)
    if not (q.is_cuda and q.dtype == torch.float32 and k.dtype == torch.float32 and v.dtype == torch.float32):
        return None
    L, N, E = q.shape
    S = k.shape[0]
    H = int(num_heads)
    if E != 32 * H or tuple(k.shape) != (S, N, E) or tuple(v.shape) != (S, N, E) or S < 32 or N * H > 65535 or L < 1:
        return None
    flags, gen = None, 0
    if isinstance(mask, DeferredMask):
        if tuple(mask.shape) != (N, L, S) or S % 4 != 0:
            return None
        if mask._bool is not None:                               # already made explicit: the bytes are the reference's tensor
            mask = mask._bool.view(torch.uint8)
        else:
            mask.check_fresh()
            mask, flags, gen = mask.mask, mask.flags, mask.gen
    elif mask is not None:
        if tuple(mask.shape) not in ((N, L, S), (N, L, (S + 3) & ~3)) or mask.dtype not in (torch.bool, torch.uint8) or not mask.is_cuda:
            return None
        mask = pad4_mask(mask) if mask.shape[-1] % 4 else mask.contiguous()      # (rows padded to dwords: see pad4_mask)
    lds_ = []
    for t in (q, k, v):
        ld = _seq_first_ld(t, N, E)
        if ld is None:
            t = t.contiguous()
            ld = E
        lds_.append((t, ld))
    (q, ldq), (k, ldk), (v, ldv) = lds_
    lib = _lib.load()
    ws = torch.empty(int(lib.univs_cross_attention_workspace(L, S, N, H)), dtype=torch.float32, device=q.device)
    out = torch.empty((L, N, E), dtype=torch.float32, device=q.device)
    with _on(q):
        rc = lib.univs_cross_attention_flagged_f32(_ptr(q), _ptr(k), _ptr(v), _ptr(mask) if mask is not None else None,
                                                   _ptr(flags) if flags is not None else None, gen, L, S, N, H, 32,
                                                   ldq, ldk, ldv, float(scale), _ptr(ws), _ptr(out), _stream_ptr(q))
    if rc == _lib.ERR_NOT_IMPLEMENTED:
        return None
    _lib.check(rc, "cross_attention")
    return out


MMA_DTYPES = {"f32": 0, "f16": 1, "f16x3": 2}      # UNIVS_MMA_F32 / _F16 / _F16X3 (include/univs_hip.h)


def window_attention_image(qkv, qkv_bias, bias, shift_mask, H, W, window_size, shift, scale, mma="f32"):
    """Swin window attention on tokens in image order: qkv [B, H*W, 3, nH, hd] (the qkv Linear applied to the
    un-padded tokens) -> [B, H*W, nH*hd]; pad / roll / window_partition / window_reverse / crop of
    swin.py:252-284 happen inside the kernel.  `qkv_bias` [3*nH*hd] or None supplies q/k/v of the padded
    pixels; `shift_mask` [nW, ws*ws, ws*ws] is required when shift > 0.  `mma`: operand precision of the two
    matrix products, "f32" (exact), "f16x3" (fp32-accurate: two fp16 parts per operand, three products) or "f16" (fp16
    operands, fp32 accumulation and softmax: BASELINE config 5)."""
    if mma not in MMA_DTYPES:
        raise ValueError(f"window_attention_image: mma={mma!r} (one of {sorted(MMA_DTYPES)})")
    _inference_only("window_attention_image", qkv, qkv_bias, bias)
    _require_gpu("window_attention_image", qkv, bias)
    if qkv.dtype != torch.float32 or qkv.dim() != 5:
        raise RuntimeError("window_attention_image: float32 [B, H*W, 3, nH, hd] only")
    B, L, three, nH, hd = qkv.shape
    ws = int(window_size)
    n = ws * ws
    if three != 3 or L != H * W or tuple(bias.shape) != (nH, n, n):
        raise RuntimeError("window_attention_image: bad shapes")
    nW = ((H + ws - 1) // ws) * ((W + ws - 1) // ws)
    if shift:
        if shift_mask is None:
            raise RuntimeError("window_attention_image: shift > 0 needs the shift mask")
        _require_gpu("window_attention_image", shift_mask)
        if tuple(shift_mask.shape) != (nW, n, n):
            raise RuntimeError("window_attention_image: bad shift_mask shape")
    else:
        shift_mask = None
    if qkv_bias is not None:
        _require_gpu("window_attention_image", qkv_bias)
        if qkv_bias.numel() != 3 * nH * hd:
            raise RuntimeError("window_attention_image: bad qkv_bias shape")
    out = torch.empty((B, L, nH * hd), dtype=torch.float32, device=qkv.device)
    with _on(qkv):
        rc = _lib.load().univs_window_attention_image_mma(
            _ptr(qkv), _ptr(qkv_bias) if qkv_bias is not None else None, _ptr(bias),
            _ptr(shift_mask) if shift_mask is not None else None, B, int(H), int(W), ws, int(shift), nH, hd,
            float(scale), MMA_DTYPES[mma], _ptr(out), _stream_ptr(qkv))
    _lib.check(rc, "window_attention_image")
    return out


def msda_prepare(proj, n_off, reference_points, spatial_shapes, num_heads, num_levels, num_points):
    """Elementwise tail of MSDeformAttn.forward (ms_deform_attn.py:100-113) in one pass: `proj` [N, Lq, C]
    holds the sampling offsets in columns [0, M*L*P*2) and the attention logits in columns [n_off, ...);
    `reference_points` [N or 1, Lq, L, 2].  Returns (sampling_locations [N,Lq,M,L,P,2], attention_weights
    [N,Lq,M,L,P]) -- the operands of `ms_deform_attn_forward`."""
    proj = proj.contiguous()
    reference_points = reference_points.contiguous()
    _inference_only("msda_prepare", proj, reference_points)
    _require_gpu("msda_prepare", proj, reference_points)
    if proj.dtype != torch.float32 or proj.dim() != 3 or reference_points.dtype != torch.float32:
        raise RuntimeError("msda_prepare: float32 proj [N, Lq, C] and reference_points only")
    N, Lq, C = proj.shape
    M, L, P = int(num_heads), int(num_levels), int(num_points)
    if tuple(reference_points.shape[1:]) != (Lq, L, 2) or reference_points.shape[0] not in (1, N):
        raise RuntimeError("msda_prepare: reference_points must be [N or 1, Lq, L, 2]")
    sh, _, L2 = _host_shapes(spatial_shapes, [0] * L)
    if L2 != L:
        raise RuntimeError("msda_prepare: spatial_shapes / num_levels mismatch")
    loc = torch.empty((N, Lq, M, L, P, 2), dtype=torch.float32, device=proj.device)
    attn = torch.empty((N, Lq, M, L, P), dtype=torch.float32, device=proj.device)
    rbs = 0 if reference_points.shape[0] == 1 else Lq * L * 2
    with _on(proj):
        rc = _lib.load().univs_msda_prepare_f32(_ptr(proj), C, int(n_off), _ptr(reference_points), rbs, sh, N, Lq, M, L, P,
                                               _ptr(loc), _ptr(attn), _stream_ptr(proj))
    _lib.check(rc, "msda_prepare")
    return loc, attn
