"""What exactly goes wrong in a kernel that runs BESIDE gemm_f16x3_tile (another stream)?  Victims: tools/probes/cohab_victim.hip, y = x * scale[r] +
bias[r] written five ways (0: hipcc's v_pk_fma_f32 with cross-half operand selection, 1: the same by hand with the affine pair loaded into
registers that held (7, 1), 2: per-component v_fma_f32, 3: as 1 with idle cycles behind the wait, 4: packed without cross-half selection).
Every wrong element is classified: x * scale (bias missing / stale 0), x * scale + 1 (the bias register still held its old value), x * 7 + bias
(stale scale), other.        python tools/race_probe8.py"""
import argparse
import ctypes
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univs_amd import ops  # noqa: E402

NAMES = {0: "0 hipcc pk_fma op_sel", 1: "1 asm pk_fma op_sel, regs held (7, 1)", 2: "2 asm v_fma_f32 x 4, regs held 7 / 1",
         3: "3 as 1 + 16 idle cycles behind the wait", 4: "4 asm pk_fma, no cross-half selection"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--set", default="base", choices=["base", "micro", "tile", "gemms", "forms"], help="aggressors: base (tile GEMM, library matmul, LDS squatters), "
                    "micro (kernels of ONE repeated instruction), tile (the tile GEMM only: run under UNIVS_HIP_LIB=<an ablation build>)")
    ap.add_argument("--variants", default="0,1,2,3,4")
    args = ap.parse_args()
    for v in list(NAMES):
        if str(v) not in args.variants.split(","):
            del NAMES[v]
    here = os.path.dirname(os.path.abspath(__file__))
    so = "/tmp/libcohab.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "probes", "cohab_victim.hip"),
                           "-o", so])
    lib = ctypes.CDLL(so)
    lib.cohab_affine.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.cohab_squat.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.cohab_pk_form.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.cohab_mfma_kind.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.cohab_one_instruction.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    g = torch.Generator().manual_seed(5)
    R, C = 1280, 14720
    x = torch.randn(R, C, generator=g).to(dev)
    aff = torch.stack([torch.rand(R, generator=g) + 0.5, torch.randn(R, generator=g) * 0.5 + 3.0], 1).contiguous().to(dev)
    sink = torch.zeros(4, device=dev)
    side = torch.cuda.Stream()

    if args.set == "forms":
        return forms(args, lib, x, aff, sink, side)

    def victim(var):
        y = torch.empty_like(x)
        rc = lib.cohab_affine(var, x.data_ptr(), aff.data_ptr(), y.data_ptr(), R, C, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        return y
    refs = {v: victim(v).clone() for v in NAMES}
    torch.cuda.synchronize()
    exact = torch.addcmul(aff[:, 1:2].double(), x.double(), aff[:, 0:1].double())
    for v in NAMES:
        print(f"variant {v}: max |y - (x * scale + bias in float64)| alone = {(refs[v].double() - exact).abs().max().item():.3e}")

    xa, wa, ba = torch.randn(5, 920, 768, device=dev), torch.randn(3072, 768, device=dev) * 0.05, torch.randn(3072, device=dev)
    big = torch.randn(4096, 4096, device=dev)
    aggressors = {
        "none": lambda: None,
        "tile GEMM 4600 x 768 -> 3072 (linear_fused)": lambda: ops.linear_fused(xa, wa, ba, act="gelu"),
        "library matmul 4096^3": lambda: big @ big,
        "LDS squatters 512 x 72 KB": lambda: lib.cohab_squat(512, 72 * 1024, 200000, sink.data_ptr(), torch.cuda.current_stream().cuda_stream),
        "LDS squatters 256 x 140 KB": lambda: lib.cohab_squat(256, 140 * 1024, 200000, sink.data_ptr(), torch.cuda.current_stream().cuda_stream),
    }
    assert aggressors["tile GEMM 4600 x 768 -> 3072 (linear_fused)"]() is not None
    micro = ["v_permlane32_swap", "v_permlane16_swap", "v_add_u32_sdwa src1_sel:BYTE_0", "v_fma_mixlo/hi_f16", "v_mfma_f32_16x16x32_f16",
             "v_pk_mul_f32 op_sel_hi:[1,0]", "v_pk_fma_f32 op_sel_hi:[1,1,0]", "v_mov_b32_dpp row_shr:1 bound_ctrl", "ds_read_b128",
             "v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,0,1]", "v_cmp_lt_i32_sdwa src0_sel:BYTE_0", "v_pk_mul_f32 neg_lo neg_hi"]
    if args.set == "micro":
        aggressors = {f"only {n}": (lambda k=k: lib.cohab_one_instruction(k, 1024, 400, sink.data_ptr(), torch.cuda.current_stream().cuda_stream))
                      for k, n in enumerate(micro)}
    elif args.set == "gemms":          # the other three-product GEMM kernels, long enough to overlap the victims
        M1 = 5 * 184 * 320
        x1, w1, b1 = torch.randn(M1, 96, device=dev), torch.randn(288, 96, device=dev) * 0.1, torch.randn(288, device=dev)
        x2 = torch.randn(96600, 256, device=dev)
        wa2, ba2, wb2, bb2 = torch.randn(1024, 256, device=dev) * 0.05, torch.randn(1024, device=dev), torch.randn(256, 1024, device=dev) * 0.03, torch.randn(256, device=dev)
        x3, w3 = torch.randn(5, 256, 184, 320, device=dev), torch.randn(256, 256, 3, 3, device=dev) * 0.02
        x4, w4, b4 = torch.randn(18400, 384, device=dev), torch.randn(1536, 384, device=dev) * 0.05, torch.randn(1536, device=dev)
        aggressors = {
            "linear_f16x3 294400 x 96 -> 288": lambda: ops.linear_fused(x1, w1, b1),
            "linear 18400 x 384 -> 1536 gelu": lambda: ops.linear_fused(x4, w4, b4, act="gelu"),
            "mlp_f16x3 96600 x 256 -> 1024 -> 256": lambda: ops.mlp_fused(x2, wa2, ba2, wb2, bb2, "relu", residual=x2),
            "gemm_f16x3_stream (3 x 3 convolution 256 -> 256 @ 184 x 320 x 5)": lambda: ops.conv3x3(x3, w3),
        }
        for n, f in aggressors.items():
            assert f() is not None, n
    elif args.set == "tile":
        from univs_amd import _lib
        print("library:", _lib.LIB_PATH)
        aggressors = {k: v for k, v in aggressors.items() if k.startswith("tile")}
    for an, f in aggressors.items():
        print(f"== beside: {an}", flush=True)
        for v, vn in NAMES.items():
            bad, kinds, lanes, comps = 0, {"x*scale": 0, "x*scale+1": 0, "x*7+bias": 0, "other": 0}, torch.zeros(4, dtype=torch.long), torch.zeros(4, dtype=torch.long)
            for it in range(args.iters):
                with torch.cuda.stream(side):
                    for _ in range(4):
                        f()
                y = victim(v)
                d = (y != refs[v]).nonzero()
                if len(d):
                    bad += 1
                    r_, c_ = d[:, 0], d[:, 1]
                    got, xs, sc, bi = y[r_, c_], x[r_, c_], aff[r_, 0], aff[r_, 1]
                    xd, sd, bd = xs.double(), sc.double(), bi.double()              # (products of two floats are exact in float64)
                    k0 = (got == (xd * sd).float())
                    k1 = (got == (xd * sd + 1.0).float()) & ~k0
                    k2 = (got == (xd * 7.0 + bd).float()) & ~k0 & ~k1
                    kinds["x*scale"] += int(k0.sum())
                    kinds["x*scale+1"] += int(k1.sum())
                    kinds["x*7+bias"] += int(k2.sum())
                    kinds["other"] += int((~k0 & ~k1 & ~k2).sum())
                    lanes += torch.bincount(((r_ % 16) % 4).cpu(), minlength=4)        # which quarter of the wave (16 lanes share a row)
                    comps += torch.bincount((c_ % 4).cpu(), minlength=4)
            torch.cuda.synchronize()
            extra = f"; wrong elements by kind {kinds}; by quarter-wave {lanes.tolist()}; by component {comps.tolist()}" if bad else ""
            print(f"   {vn}: {bad} of {args.iters} runs differ{extra}", flush=True)


PK_FORMS = {      # name, expected (low, high) from a = (a0, a1), b = (b0, b1), and what the low / high result would be with the cross-half operand read as 0
    0: ("v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,1]  (low <- src1.hi)", lambda a0, a1, b0, b1: (a0 * b1, a1 * b1)),
    1: ("v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[1,1]  (low <- src0.hi)", lambda a0, a1, b0, b1: (a1 * b0, a1 * b1)),
    2: ("v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,1]  (low <- src1.hi)", lambda a0, a1, b0, b1: (a0 + b1, a1 + b1)),
    3: ("v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,1,0]  (low <- src1.hi, high <- src2.lo)", lambda a0, a1, b0, b1: (a0 * b1 + b0, a1 * b1 + b0)),
    4: ("v_pk_fma_f32 op_sel:[1,0,0] op_sel_hi:[1,0,1]  (low <- src0.hi, high <- src1.lo)", lambda a0, a1, b0, b1: (a1 * b0 + b0, a1 * b0 + b1)),
    5: ("v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,0,1]  (low <- src2.hi, high <- src1.lo): the form of the transposes", lambda a0, a1, b0, b1: (a0 * b0 + b1, a1 * b0 + b1)),
    6: ("v_pk_fma_f32 op_sel:[0,0,0] op_sel_hi:[1,1,0]  (high <- src2.lo)", lambda a0, a1, b0, b1: (a0 * b0 + b0, a1 * b1 + b0)),
    7: ("v_pk_mul_f32 op_sel:[0,0] op_sel_hi:[1,0]  (high <- src1.lo)", lambda a0, a1, b0, b1: (a0 * b0, a1 * b0)),
}
MFMA_KINDS = ["v_mfma_f32_16x16x32_f16, one dependent chain", "v_mfma_f32_16x16x32_f16, two chains", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_16x16x32_bf16",
              "v_mfma_f32_16x16x16_f16", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_16x16x32_f16 with 16 idle cycles between"]


def forms(args, lib, x, aff, sink, side):
    """Every cross-half operand selection of the packed f32 instructions beside every kind of MFMA wave."""
    R, C = x.shape
    a0, a1 = x[:, 0::2].double(), x[:, 1::2].double()
    b0, b1 = aff[:, 0:1].double(), aff[:, 1:2].double()

    def run(form):
        y = torch.empty_like(x)
        rc = lib.cohab_pk_form(form, x.data_ptr(), aff.data_ptr(), y.data_ptr(), R, C, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        return y
    refs = {}
    for f_, (name, fn) in PK_FORMS.items():
        refs[f_] = run(f_).clone()
        lo, hi = fn(a0, a1, b0, b1)
        exp = torch.stack([lo, hi], -1).flatten(1).float()
        print(f"form {f_} alone: {int((refs[f_] != exp).sum())} elements differ from the expected values   [{name}]")
    torch.cuda.synchronize()
    for k, kn in enumerate(MFMA_KINDS):
        print(f"== beside waves of {kn}", flush=True)
        for f_, (name, fn) in PK_FORMS.items():
            if k > 0 and f_ not in (0, 1, 5):
                continue
            bad, n_wrong, quarter, half, zero_like = 0, 0, torch.zeros(4, dtype=torch.long), torch.zeros(2, dtype=torch.long), 0
            for it in range(args.iters):
                with torch.cuda.stream(side):
                    for _ in range(2):
                        lib.cohab_mfma_kind(k, 1024, 300, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
                y = run(f_)
                d = (y != refs[f_]).nonzero()
                if len(d):
                    bad += 1
                    n_wrong += len(d)
                    quarter += torch.bincount(((d[:, 0] % 16) % 4).cpu(), minlength=4)
                    half += torch.bincount((d[:, 1] % 2).cpu(), minlength=2)
            torch.cuda.synchronize()
            extra = f"; {n_wrong} wrong elements; by quarter-wave {quarter.tolist()}; low / high result {half.tolist()}" if bad else ""
            print(f"   form {f_}: {bad} of {args.iters} runs differ{extra}", flush=True)


if __name__ == "__main__":
    main()
