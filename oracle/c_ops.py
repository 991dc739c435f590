"""numpy/ctypes binding of oracle/ops_ref.c (TEST INFRASTRUCTURE; see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "ops_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(src) > os.path.getmtime(_SO):
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def msda_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight):
    """value [N,S,M,D], loc [N,Lq,M,L,P,2], attn [N,Lq,M,L,P] -> [N,Lq,M*D] (float32 or float64)."""
    dt = value.dtype
    assert dt in (np.float32, np.float64)
    value = np.ascontiguousarray(value)
    loc = np.ascontiguousarray(sampling_loc, dtype=dt)
    attn = np.ascontiguousarray(attn_weight, dtype=dt)
    sh = np.ascontiguousarray(spatial_shapes, dtype=np.int64).reshape(-1)
    st = np.ascontiguousarray(level_start_index, dtype=np.int64).reshape(-1)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = np.empty((N, Lq, M * D), dtype=dt)
    fn = lib().oracle_msda_forward_f32 if dt == np.float32 else lib().oracle_msda_forward_f64
    fn(_p(value), _p(sh), _p(st), _p(loc), _p(attn), N, S, M, D, L, Lq, P, _p(out))
    return out


def mask_decode(mask_embed, mask_features):
    """mask_embed [T,Q,C], mask_features [T,C,H,W] -> [Q,T,H,W]."""
    e = np.ascontiguousarray(mask_embed, dtype=np.float32)
    f = np.ascontiguousarray(mask_features, dtype=np.float32)
    T, Q, C = e.shape
    _, _, H, W = f.shape
    out = np.empty((Q, T, H, W), dtype=np.float32)
    lib().oracle_mask_decode_f32(_p(e), _p(f), T, Q, C, ctypes.c_longlong(H * W), _p(out))
    return out


def attn_mask_from_logits(logits):
    """logits [T,Q,hw] -> bool [T,Q,hw] with the all-masked-row reset."""
    x = np.ascontiguousarray(logits, dtype=np.float32)
    T, Q, hw = x.shape
    m = np.empty((T, Q, hw), dtype=np.uint8)
    lib().oracle_attn_mask_from_logits(_p(x), T, Q, ctypes.c_longlong(hw), _p(m))
    return m.astype(bool)


def window_attention(qkv, bias, shift_mask, scale):
    """qkv [B_,Ntok,3,nH,hd], bias [nH,Ntok,Ntok], shift_mask [nW,Ntok,Ntok] or None -> [B_,Ntok,nH*hd]."""
    qkv = np.ascontiguousarray(qkv, dtype=np.float32)
    bias = np.ascontiguousarray(bias, dtype=np.float32)
    B_, Ntok, _, nH, hd = qkv.shape
    nW = 1
    mp = None
    if shift_mask is not None:
        shift_mask = np.ascontiguousarray(shift_mask, dtype=np.float32)
        nW = shift_mask.shape[0]
        mp = _p(shift_mask)
    out = np.empty((B_, Ntok, nH * hd), dtype=np.float32)
    lib().oracle_window_attention_f32(_p(qkv), _p(bias), mp, B_, nW, Ntok, nH, hd, ctypes.c_float(scale), _p(out))
    return out


def rle_encode(mask):
    """mask [H, W] (bool / 0-1) -> (counts int64 array, compressed COCO RLE string) by the C restatement of maskApi.c."""
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    h, w = m.shape
    out = ctypes.create_string_buffer(7 * (h * w + 1) + 1)
    counts = np.empty(h * w + 1, dtype=np.int64)
    n = ctypes.c_longlong(0)
    fn = lib().oracle_rle_encode
    fn.restype = ctypes.c_longlong
    fn(_p(m), h, w, out, _p(counts), ctypes.byref(n))
    return counts[: n.value].copy(), out.value.decode("ascii")
