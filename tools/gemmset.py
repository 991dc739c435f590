"""The three-product GEMM launches of one config-2 clip, one by one (run on the GPU box): shape, time, fp16-MFMA share and HBM
share.  With UNIVS_HIP_LIB=univs_amd/libunivs_hip_<nosplit|nomfma|nosplit_nomfma>.so the same launches run an instrumented
build (csrc/f16x3.h; results wrong, timing only): what the x-operand split / the matrix instructions cost in place.
    python tools/gemmset.py [--tag default]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univs_amd import ops, synth  # noqa: E402
from tools.kbench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default=os.path.basename(os.environ.get("UNIVS_HIP_LIB", "default")))
    ap.add_argument("--T", type=int, default=5)
    ap.add_argument("--linear-ablate", type=int, default=0, help="UnivsConfig.linear_ablate (5: no XCD-aware (row range, pass) order; 10: the phase-shifted fused MLP)")
    args = ap.parse_args()
    if args.linear_ablate:
        ops.configure(linear_ablate=args.linear_ablate)
    dev = torch.device("cuda:0")
    T = args.T
    S = 19320
    rows = []

    def lin(name, M, K, N, act=None, res=False, blocked=None, count=1):
        x = synth.normal(f"gs/x/{M}x{K}", (M, K)).to(dev)
        w = synth.normal(f"gs/w/{N}x{K}", (N, K), std=K ** -0.5).to(dev)
        b = synth.normal(f"gs/b/{N}", (N,)).to(dev)
        r = synth.normal(f"gs/r/{M}x{N}", (M, N)).to(dev) if res else None
        if blocked:
            xb = x.view(T, M // T, K)
            fn = lambda: ops.linear_blocked(xb, w, b, M // T, blocked)
        else:
            fn = lambda: ops.linear_fused(x, w, b, act=act, residual=r)
        if fn() is None:
            rows.append(dict(name=name, us=None))
            return
        t = timeit(fn, iters=30, warmup=5)
        fl = 2.0 * M * K * N
        by = 4.0 * M * (K + N + (N if res else 0))
        rows.append(dict(name=name, M=M, K=K, N=N, us=round(t * 1e6, 1), per_clip=count, mfma_frac=round(3 * fl / t / 2.5e15, 3),
                         hbm_frac=round(by / t / 8e12, 3)))

    # encoder layer (x 6): value_proj, merged offset / logit projection (blocked, head-major), output_proj (+ residual)
    lin("enc_value_proj_blocked16", T * S, 256, 256, blocked=16, count=6)
    lin("enc_offs_proj_blocked36", T * S, 256, 288, blocked=36, count=6)
    lin("enc_output_proj_res", T * S, 256, 256, res=True, count=6)
    # Swin stage 1 / 2 (x 2 blocks each): qkv, proj + shortcut
    lin("s1_qkv", T * 184 * 320, 96, 288, count=2)
    lin("s1_proj_res", T * 184 * 320, 96, 96, res=True, count=2)
    lin("s2_qkv", T * 92 * 160, 192, 576, count=2)
    lin("s2_proj_res", T * 92 * 160, 192, 192, res=True, count=2)
    # Swin stage 3 (x 6) and 4 (x 2)
    lin("s3_qkv", T * 3680, 384, 1152, count=6)
    lin("s3_proj_res", T * 3680, 384, 384, res=True, count=6)
    lin("s3_fc1_gelu", T * 3680, 384, 1536, act="gelu", count=6)
    lin("s3_fc2_res", T * 3680, 1536, 384, res=True, count=6)
    lin("s4_qkv", T * 920, 768, 2304, count=2)
    lin("s4_proj_res", T * 920, 768, 768, res=True, count=2)
    lin("s4_fc1_gelu", T * 920, 768, 3072, act="gelu", count=2)
    lin("s4_fc2_res", T * 920, 3072, 768, res=True, count=2)
    # patch merging reductions
    lin("merge1_384x192", T * 92 * 160, 384, 192)
    lin("merge2_768x384", T * 3680, 768, 384)
    lin("merge3_1536x768", T * 920, 1536, 768)
    # decoder K / V projections of the three levels (3 layers each: N = 768)
    lin("dec_kv_l8", T * 14720, 256, 768, count=2)
    lin("dec_kv_l16", T * 3680, 256, 768, count=2)
    lin("dec_kv_l32", T * 920, 256, 768, count=2)

    # fused MLPs
    for nm, Mr, C, Hd, act, with_res, cnt in (("enc_ffn", T * S, 256, 1024, "relu", False, 6), ("s1_mlp", T * 184 * 320, 96, 384, "gelu", True, 2),
                                              ("s2_mlp", T * 92 * 160, 192, 768, "gelu", True, 2)):
        xs = synth.normal(f"gs/mlp/x{C}", (Mr, C)).to(dev)
        w1 = synth.normal(f"gs/mlp/w1/{C}", (Hd, C), std=C ** -0.5).to(dev)
        b1 = synth.normal(f"gs/mlp/b1/{C}", (Hd,), std=0.5).to(dev)
        w2 = synth.normal(f"gs/mlp/w2/{C}", (C, Hd), std=Hd ** -0.5).to(dev)
        b2 = synth.normal(f"gs/mlp/b2/{C}", (C,), std=0.5).to(dev)
        rs = synth.normal(f"gs/mlp/r{C}", (Mr, C)).to(dev) if with_res else None
        t = timeit(lambda: ops.mlp_fused(xs, w1, b1, w2, b2, act, residual=rs), iters=20, warmup=4)
        fl = 4.0 * Mr * C * Hd
        rows.append(dict(name=nm, M=Mr, K=C, N=Hd, us=round(t * 1e6, 1), per_clip=cnt, mfma_frac=round(3 * fl / t / 2.5e15, 3),
                         hbm_frac=round(4.0 * Mr * C * (3 if with_res else 2) / t / 8e12, 3)))
    # convolutions of the FPN
    xc = synth.normal("gs/conv/x", (T, 256, 184, 320)).to(dev)
    wc = synth.normal("gs/conv/w3", (256, 256, 3, 3), std=1 / 48).to(dev)
    t = timeit(lambda: ops.conv3x3(xc, wc), iters=10, warmup=3)
    fl = 2.0 * T * 184 * 320 * 256 * 2304
    rows.append(dict(name="fpn_conv3x3", us=round(t * 1e6, 1), per_clip=1, mfma_frac=round(3 * fl / t / 2.5e15, 3), hbm_frac=round(2 * xc.numel() * 4 / t / 8e12, 3)))
    xn = xc.permute(0, 2, 3, 1).contiguous()
    if hasattr(ops, "conv3x3_nhwc") and ops.conv3x3_nhwc(xn, wc) is not None:
        same = bool(torch.equal(ops.conv3x3_nhwc(xn, wc), ops.conv3x3(xc, wc)))
        t = timeit(lambda: ops.conv3x3_nhwc(xn, wc), iters=10, warmup=3)
        rows.append(dict(name="fpn_conv3x3_nhwc_input", us=round(t * 1e6, 1), per_clip=0, mfma_frac=round(3 * fl / t / 2.5e15, 3), bit_identical_to_nchw=same))
    del xn
    w1 = synth.normal("gs/conv/w1", (256, 256, 1, 1), std=1 / 16).to(dev)
    b1 = synth.normal("gs/conv/b1", (256,)).to(dev)
    t = timeit(lambda: ops.conv1x1(xc, w1, b1), iters=10, warmup=3)
    fl = 2.0 * T * 184 * 320 * 256 * 256
    rows.append(dict(name="mask_features_conv1x1", us=round(t * 1e6, 1), per_clip=1, mfma_frac=round(3 * fl / t / 2.5e15, 3), hbm_frac=round(2 * xc.numel() * 4 / t / 8e12, 3)))
    xl = synth.normal("gs/conv/xl", (T, 96, 184, 320)).to(dev)
    wl = synth.normal("gs/conv/wl", (256, 96, 1, 1), std=0.1).to(dev)
    t = timeit(lambda: ops.conv1x1(xl, wl, None), iters=10, warmup=3)
    rows.append(dict(name="lateral_conv1x1_96", us=round(t * 1e6, 1), per_clip=1, hbm_frac=round((xl.numel() + xc.numel()) * 4 / t / 8e12, 3)))
    total = sum(r["us"] * r.get("per_clip", 1) for r in rows if r.get("us"))
    for r in rows:
        print(args.tag, json.dumps(r))
    print(args.tag, json.dumps(dict(name="TOTAL_per_clip_ms", ms=round(total / 1e3, 3))))


if __name__ == "__main__":
    main()
