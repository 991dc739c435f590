"""Python-side switches of the hot path (which operator implementation a module routes to).  ONE process-wide object,
`SWITCHES`; its defaults are read from the environment ONCE, at import (so a launcher can still A/B with `UNIVS_*=...
python bench.py`), and tests / tools change it programmatically:

    from univs_amd.switches import SWITCHES, override
    with override(msda_strips=False): ...

The native library has its own settings object (include/univs_hip.h: UnivsConfig, `ops.configure`) and reads no environment
variable at all.
"""
import contextlib
import dataclasses
import os


def _flag(name, default):
    v = os.environ.get(name)
    return default if v is None or v == "" else v != "0"


@dataclasses.dataclass
class Switches:
    # the encoder's MSDeformAttn core on head-major operands (csrc/msda_strips.hip); False: msda_prepare + the standard-layout
    # operator (msda_tiled2.hip / generic)
    msda_strips: bool = True
    # ... its sixth generation (csrc/msda_heads.hip: a lane owns a sample of a FULL head, one workgroup per CU, lockstep column
    # segments); False: generation 5 (csrc/msda_strips.hip)
    msda_heads: bool = True
    # token Linears on the hand-written fp32-accurate GEMM kernels (False: library GEMMs)
    split_linear: bool = True
    # the 3 x 3 FPN output convolution on the three-product fp16 kernel (False: MIOpen)
    split_conv: bool = True
    # Swin qkv / proj / fc1+GELU / fc2+shortcut on the fused hand-written Linears (False: library GEMM + elementwise passes);
    # `swin_fused_parts`: diagnostic bit set, 1 qkv / proj, 2 fc1 + GELU, 4 fc2 + shortcut
    swin_fused_linear: bool = True
    swin_fused_parts: int = 7
    # two-Linear MLPs (encoder FFN, Swin Mlp + shortcut at C <= 256) in ONE kernel, hidden activations in registers
    # (csrc/mlp_f16x3.hip); False: two fused Linears
    fused_mlp: bool = True
    # widest C the Swin blocks route to the fused MLP.  The kernel also covers C = 384 (one weight image, two barriers per chunk,
    # 4-wave workgroups) but measured 321 us against 257 us for the two fused Linears at Swin stage 3 (18 400 rows: 288 row groups
    # on 256 CUs, one wave per SIMD) -- profiles/r04_kbench_mlp_v2.txt
    fused_mlp_max_c: int = 256
    # encoder layer: norm1 evaluated inside the fused FFN kernel (on the x tile; its result is also the FFN's residual) instead of a
    # LayerNorm launch of its own
    fused_norm1: bool = True
    # the decoder's per-token Linears (a few hundred to a few thousand rows) with `tgt + query_pos` in front and `norm(tgt + .)` behind
    # in one launch each (csrc/small_linear.hip); False: library GEMMs + elementwise / LayerNorm launches
    small_linear: bool = True
    # decoder cross-attention core (scores, mask, softmax, P V) as one pass over the keys (csrc/cross_attn.hip); False: two
    # library GEMMs around the masked-softmax kernel
    fused_cross_attention: bool = True
    # widest K routed to the hand-written Linears
    linear_kmax: int = 4096
    # Linears with K >= presplit_kmin and the 3 x 3 convolution run on the three-product fp16 kernel with weights split once
    # per tensor (csrc/gemm_f16x3_stream.hip; ops.presplit_weights caches the split); 0: only the W-resident kernels (K <= 768), which
    # split W in every workgroup; wider Linears and the convolution then go to the libraries
    presplit_kmin: int = 768
    # few-rows MLPs of 256-channel Linears (the mask-embedding MLP of every prediction head) as ONE launch, the hidden rows handed from
    # stage to stage through LDS (csrc/small_linear.hip: small_chain_kernel; bit-identical to the separate launches); `small_mlp_norm`:
    # the `decoder_norm` LayerNorm in front of it inside the same launch (its own fp32 rounding: not bit-identical to ATen's kernel)
    small_mlp_chain: bool = True
    # ProCA (a prompt query attends to its own prompt tokens) without the per-layer concatenations / transpositions: the dense tokens'
    # K / V projections on the tokens where they lie, q / k0 / v0 in one few-rows launch, one attention launch (csrc/proca_attn.hip)
    fused_proca: bool = True
    # the visual-prompt sampler of a prompted clip (candidate pixels, draws -> pixels, dense tokens, cross-attention masks) as a handful
    # of kernels instead of ~250 ATen launches (csrc/prompt_sampler.hip; bit-identical to the ATen formulation, which stays the CPU path)
    fused_sampler: bool = True
    small_mlp_norm: bool = True
    # the W-resident Linears (K <= 768) stage their slab of W from the split image cached per weight tensor (a copy) instead of
    # splitting it in every workgroup of every launch (13 - 15 us per launch: profiles/r05_gemm_phase_trace_v1.txt)
    resident_presplit: bool = True
    # prompt sampler draws: "reference" (the reference's host-side randperm calls in its order: its random stream draw for draw; on a
    # 720p video with a large entity ~70 ms of host time per clip), "device" (the same distributions drawn by the device generator, no
    # host round trip), or "auto" (default): "device" for GPU tensors, "reference" on the CPU.  Parity runs against recorded reference
    # states set "reference" (tests/conftest.py does, through UNIVS_SAMPLER) or replay the reference's draws.
    sampler: str = "auto"
    # hipGraph replay of the static parts of a clip (backbone, pixel decoder): see univs_amd/graphs.py
    graphs: bool = False


SWITCHES = Switches(
    msda_strips=_flag("UNIVS_MSDA_STRIPS", True), msda_heads=_flag("UNIVS_MSDA_HEADS", True), split_linear=_flag("UNIVS_SPLIT_LINEAR", True),
    split_conv=_flag("UNIVS_SPLIT_CONV", True), swin_fused_linear=_flag("UNIVS_SWIN_FUSED_LINEAR", True),
    swin_fused_parts=int(os.environ.get("UNIVS_SWIN_FUSED_PARTS", "7")), linear_kmax=int(os.environ.get("UNIVS_LINEAR_KMAX", "4096")),
    sampler=os.environ.get("UNIVS_SAMPLER", "auto"), graphs=_flag("UNIVS_GRAPHS", False),
    presplit_kmin=int(os.environ.get("UNIVS_PRESPLIT_KMIN", "768")), fused_mlp=_flag("UNIVS_FUSED_MLP", True),
    fused_cross_attention=_flag("UNIVS_FUSED_XATTN", True), fused_norm1=_flag("UNIVS_FUSED_NORM1", True), small_linear=_flag("UNIVS_SMALL_LINEAR", True),
    resident_presplit=_flag("UNIVS_RESIDENT_PRESPLIT", True), small_mlp_chain=_flag("UNIVS_SMALL_MLP_CHAIN", True), fused_proca=_flag("UNIVS_FUSED_PROCA", True),
    fused_sampler=_flag("UNIVS_FUSED_SAMPLER", True), small_mlp_norm=_flag("UNIVS_SMALL_MLP_NORM", True))
if SWITCHES.sampler not in ("auto", "reference", "device"):
    raise ValueError(f"UNIVS_SAMPLER={SWITCHES.sampler!r} (expected 'auto', 'reference' or 'device')")


@contextlib.contextmanager
def override(**kw):
    old = {k: getattr(SWITCHES, k) for k in kw}
    for k, v in kw.items():
        if not hasattr(SWITCHES, k):
            raise KeyError(k)
        setattr(SWITCHES, k, v)
    try:
        yield SWITCHES
    finally:
        for k, v in old.items():
            setattr(SWITCHES, k, v)
