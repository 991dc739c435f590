"""ctypes binding of libunivs_hip.so (the C ABI declared in include/univs_hip.h).

No fallback: if the shared library is missing or a symbol cannot be resolved this raises.  The
product path never routes through `oracle/` or a CPU implementation.
"""
import ctypes
import os

# Load order matters: PyTorch-ROCm bundles its own libamdhip64 and must bring the HIP runtime into the
# process FIRST.  If libunivs_hip.so (linked against the system ROCm) is dlopen'ed before torch, two
# runtime instances coexist and launches on torch's streams fail with "no ROCm-capable device".
import torch  # noqa: F401  (kept first on purpose)

_HERE = os.path.dirname(os.path.abspath(__file__))
# UNIVS_HIP_LIB: debug override (tools/msda_trace.py loads an instrumented build of the same sources)
LIB_PATH = os.environ.get("UNIVS_HIP_LIB") or os.path.join(_HERE, "libunivs_hip.so")

OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_NOT_IMPLEMENTED = -2
ERR_LAUNCH = -3

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int

# name -> (restype, argtypes); must list every symbol of include/univs_hip.h (tests check this)
SIGNATURES = {
    "univs_version": (_c.c_char_p, []),
    "univs_last_error": (_c.c_char_p, []),
    "univs_msda_forward_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "univs_msda_forward_f64": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "univs_msda_backward_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "univs_msda_forward_strips_f32": (_I, [_P, _P, _P, _P, _P, _c.c_longlong, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "univs_msda_forward_heads_f32": (_I, [_P, _P, _P, _P, _P, _c.c_longlong, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "univs_linear_blocked_f32": (_I, [_P, _P, _P, _c.c_longlong, _I, _I, _I, _I, _P, _P]),
    "univs_configure": (_I, [_P]),
    "univs_get_config": (_I, [_P]),
    "univs_msda_set_impl": (_I, [_I]),
    "univs_msda_last_impl": (_I, []),
    "univs_msda_last_tiled_generation": (_I, []),
    "univs_transpose_f32": (_I, [_P, _c.c_longlong, _I, _I, _P, _P]),
    "univs_transpose_strided_f32": (_I, [_P, _c.c_longlong, _I, _I, _c.c_longlong, _P, _P]),
    "univs_transpose_ex_f32": (_I, [_P, _c.c_longlong, _I, _I, _c.c_longlong, _P, _P, _c.c_longlong, _P, _P, _P]),
    "univs_linear_fused_f32": (_I, [_P, _P, _P, _P, _c.c_longlong, _I, _I, _I, _P, _P]),
    "univs_mask_decode_set_impl": (_I, [_I]),
    "univs_mask_decode_last_impl": (_I, []),
    "univs_mask_decode_f32": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "univs_mask_decode_attn_f32": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "univs_window_attention_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _c.c_float, _P, _P]),
    "univs_msda_prepare_f32": (_I, [_P, _I, _I, _P, _c.c_longlong, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "univs_window_attention_image_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _c.c_float, _P, _P]),
    "univs_presplit_weights_f32": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "univs_linear_presplit_f32": (_I, [_P, _P, _P, _P, _P, _c.c_longlong, _I, _I, _I, _P, _P]),
    "univs_conv3x3_presplit_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "univs_conv3x3_nhwc_presplit_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "univs_linear_resident_presplit_f32": (_I, [_P, _P, _P, _P, _P, _c.c_longlong, _I, _I, _I, _P, _P]),
    "univs_linear_blocked_presplit_f32": (_I, [_P, _P, _P, _P, _c.c_longlong, _I, _I, _I, _I, _P, _P]),
    "univs_conv1x1_presplit_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "univs_patch_embed4_f32": (_I, [_P, _P, _P, _P, _P, _c.c_float, _I, _I, _I, _I, _P, _P]),
    "univs_decoder_memory_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "univs_cross_attention_workspace": (_c.c_longlong, [_I, _I, _I, _I]),
    "univs_cross_attention_flagged_f32": (_I, [_P, _P, _P, _P, _P, _c.c_uint32, _I, _I, _I, _I, _I, _I, _I, _I, _c.c_float, _P, _P, _P]),
    "univs_mask_decode_attn_deferred_f32": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _c.c_uint32, _P]),
    "univs_attn_mask_rows_reset": (_I, [_P, _P, _c.c_uint32, _c.c_longlong, _c.c_longlong, _P]),
    "univs_cross_attention_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _c.c_float, _P, _P, _P]),
    "univs_small_linear_presplit_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _c.c_float, _c.c_longlong, _I, _I, _I, _I, _I, _P, _P]),
    "univs_small_mlp_presplit_f32": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _c.c_float, _P, _c.c_longlong, _I, _P, _P]),
    "univs_mlp_presplit_v2_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _c.c_float, _P, _P, _c.c_float, _P, _c.c_longlong, _P,
                                       _c.c_longlong, _I, _I, _I, _P, _P]),
    "univs_mlp_presplit_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _c.c_float, _P, _P, _c.c_float, _P, _c.c_longlong, _P,
                                    _c.c_longlong, _I, _I, _I, _P, _P]),
    "univs_window_attention_image_mma": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _c.c_float, _I, _P, _P]),
    "univs_upsample2x_add_f32": (_I, [_P, _P, _P, _P, _c.c_longlong, _I, _I, _P]),
    "univs_group_norm_affine_f32": (_I, [_P, _P, _P, _I, _I, _c.c_longlong, _I, _c.c_float, _P, _c.c_longlong, _P, _P]),
    "univs_bilinear_pyramid3_f32": (_I, [_P, _c.c_longlong, _I, _I, _P, _P, _P, _P]),
    "univs_bilinear_resample_f32": (_I, [_P, _P, _P, _c.c_longlong, _I, _I, _I, _I, _P]),
    "univs_normalize_pad_f32": (_I, [_P, _P, _P, _c.c_longlong, _I, _I, _I, _I, _I, _P, _P]),
    "univs_patch_merge_norm_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _c.c_float, _P, _P]),
    "univs_layer_norm_f32": (_I, [_P, _P, _P, _P, _c.c_longlong, _I, _c.c_float, _P, _P, _P]),
    "univs_layer_norm_add_f32": (_I, [_P, _P, _P, _P, _P, _c.c_longlong, _c.c_longlong, _I, _c.c_float, _P, _P, _P, _P]),
    "univs_group_norm_f32": (_I, [_P, _P, _P, _I, _I, _c.c_longlong, _I, _c.c_float, _I, _P, _c.c_longlong, _P, _P]),
    "univs_masked_softmax_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "univs_proca_attention_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _c.c_float, _P, _P]),
    "univs_prompt_prefix_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _c.c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "univs_prompt_draw": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "univs_prompt_point_pe_f32": (_I, [_P, _P, _P, _P, _c.c_float, _I, _I, _I, _P, _P]),
    "univs_token_mean_f32": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "univs_mask_stats_f32": (_I, [_P, _c.c_longlong, _I, _I, _I, _I, _c.c_float, _c.c_float, _c.c_float, _P, _P]),
    "univs_mask_stats_strided_f32": (_I, [_P, _c.c_longlong, _I, _c.c_longlong, _c.c_longlong, _I, _I, _I, _I, _c.c_float, _c.c_float, _c.c_float, _P, _P]),
    "univs_prompt_tokens_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
}

_lib = None


class UnivsHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises if the library is absent -- build it with
    `python -m univs_amd.build` (or `__graft_entry__.build()`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UnivsHipError(
            f"{LIB_PATH} not found: the HIP extension is mandatory (no CPU fallback). "
            "Build it with `python -m univs_amd.build`.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing -> loud
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc == OK:
        return
    msg = load().univs_last_error().decode("utf-8", "replace")
    if rc == ERR_NOT_IMPLEMENTED:
        raise NotImplementedError(f"{what}: {msg}")
    raise UnivsHipError(f"{what} failed (code {rc}): {msg}")
