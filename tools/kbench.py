"""Kernel micro-benchmarks at the BASELINE config-2 geometry (run on the GPU box).
Prints one line per kernel: average launch time (HIP events on the launch stream), achieved algorithmic
GB/s (SURVEY.md section 8d figures) and fraction of the 8 TB/s HBM peak."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases  # noqa: E402  (development tool: the tests' input builders)
from univs_amd import ops, synth  # noqa: E402

HBM_PEAK = 8.0e12


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=5)
    ap.add_argument("--locs", default="local", choices=["local", "uniform", "far"])
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    res = {}
    T = args.T
    shapes = [(23, 40), (46, 80), (92, 160)]
    case = dict(name="kb", shapes=shapes, N=T, M=8, D=32, P=4, encoder=True, far=(args.locs == "far"))
    value, shapes, lsi, loc, attn = cases.msda_inputs(case)
    if args.locs == "uniform":
        loc = synth.uniform("kb/uloc", tuple(loc.shape), 0.0, 1.0)
    value, loc, attn = value.to(dev), loc.to(dev), attn.to(dev)
    S = value.shape[1]
    alg = 3200.0 * S * T
    only_strips = args.only == "strips"
    if not args.only or "msda" in args.only or only_strips:
        with ops.configured(msda_impl=1):
            t = timeit(lambda: ops.ms_deform_attn_forward(value, shapes, lsi, loc, attn))
            res["msda_generic"] = dict(ms=t * 1e3, us_per_frame=t * 1e6 / T, GBps=alg / t / 1e9, frac_hbm=alg / t / HBM_PEAK)
            ref = ops.ms_deform_attn_forward(value, shapes, lsi, loc, attn)
        with ops.configured(msda_impl=2):
            t = timeit(lambda: ops.ms_deform_attn_forward(value, shapes, lsi, loc, attn))
            out = ops.ms_deform_attn_forward(value, shapes, lsi, loc, attn)
            d = (out - ref).abs()
            res["msda_tiled2"] = dict(ms=t * 1e3, us_per_frame=t * 1e6 / T, GBps=alg / t / 1e9, frac_hbm=alg / t / HBM_PEAK,
                                      gen=ops.msda_last_tiled_generation(), max_abs_diff_vs_generic=d.max().item())
        # raw projections that reproduce the benchmark's sampling locations: offsets in pixels of the target level relative to
        # the query's own pixel centre (the encoder's reference points), logits whose softmax is `attn`
        M_, L_, P_ = 8, 3, 4
        refs = []
        for (h, w) in shapes:
            ys = (torch.arange(h, dtype=torch.float32) + 0.5) / h
            xs = (torch.arange(w, dtype=torch.float32) + 0.5) / w
            yy, xx = torch.meshgrid(ys, xs, indexing="ij")
            refs.append(torch.stack([xx.reshape(-1), yy.reshape(-1)], -1))
        refp = torch.cat(refs, 0).view(1, S, 1, 2).expand(1, S, L_, 2).contiguous().to(dev)
        norm = torch.tensor([[w, h] for (h, w) in shapes], dtype=torch.float32, device=dev).view(1, 1, 1, L_, 1, 2)
        off = (loc - refp.view(1, S, 1, L_, 1, 2)) * norm
        proj = torch.cat([off.reshape(T, S, -1), attn.clamp_min(1e-30).log().reshape(T, S, -1)], -1).contiguous()
        n_off = M_ * L_ * P_ * 2
        t = timeit(lambda: ops.msda_prepare(proj, n_off, refp, shapes, M_, L_, P_))
        res["msda_prepare"] = dict(ms=t * 1e3)
        # generation 5: strips on head-major operands (half a head per workgroup, two workgroups per CU)
        vhm, qhm = ops.msda_pack_head_major(value, proj, n_off, shapes, P_)
        refq = refp[:, :, 0].contiguous()
        l_, a_ = ops.msda_prepare(proj, n_off, refp, shapes, M_, L_, P_)
        with ops.configured(msda_impl=1):
            want = ops.ms_deform_attn_forward(value, shapes, lsi, l_, a_)
        variants = (("", {}),) if only_strips else (("", {}), ("_again", {}), ("_th6", dict(msda_strip_h=6)), ("_w8", dict(msda_strip_w=8)),
                                                     ("_grid256", dict(msda_grid=256)), ("_grid512", dict(msda_grid=512)),
                                                     ("_grid768", dict(msda_grid=768)), ("_grid2048", dict(msda_grid=2048)))
        for tag, cfg in variants:
            with ops.configured(**cfg):
                got = ops.msda_forward_strips(vhm, qhm, refq, shapes, lsi, M_, P_)
                if got is not None:
                    t = timeit(lambda: ops.msda_forward_strips(vhm, qhm, refq, shapes, lsi, M_, P_))
                    res["msda_strips" + tag] = dict(ms=t * 1e3, GBps=alg / t / 1e9, frac_hbm=alg / t / HBM_PEAK,
                                                    gen=ops.msda_last_tiled_generation(),
                                                    max_abs_diff_vs_generic=(got - want).abs().max().item())
        # generation 6: a full head per lane-sample, one workgroup per CU, lockstep column segments
        vh6, qh6 = ops.msda_pack_heads(value, proj, n_off, shapes, P_)
        variants6 = (("", {}), ("_contig", dict(msda_sched=1))) if only_strips else (
            ("", {}), ("_again", {}), ("_contig", dict(msda_sched=1)), ("_w16h6", dict(msda_strip_w=16, msda_strip_h=6)),
            ("_w16h6_contig", dict(msda_strip_w=16, msda_strip_h=6, msda_sched=1)), ("_th6", dict(msda_strip_h=6)),
            ("_w8", dict(msda_strip_w=8)), ("_r5", dict(msda_halo=5)), ("_grid512", dict(msda_grid=512)), ("_grid248", dict(msda_grid=248)))
        for tag, cfg in variants6:
            with ops.configured(**cfg):
                got = ops.msda_forward_heads(vh6, qh6, refq, shapes, lsi, M_, P_)
                if got is not None:
                    t = timeit(lambda: ops.msda_forward_heads(vh6, qh6, refq, shapes, lsi, M_, P_))
                    res["msda_heads" + tag] = dict(ms=t * 1e3, GBps=alg / t / 1e9, frac_hbm=alg / t / HBM_PEAK,
                                                   gen=ops.msda_last_tiled_generation(),
                                                   max_abs_diff_vs_generic=(got - want).abs().max().item())
        x_tok = synth.normal("kb/tok", (T, S, 256)).to(dev)
        w_v, b_v = synth.normal("kb/wv", (256, 256), std=0.05).to(dev), synth.normal("kb/bv", (256,)).to(dev)
        w_q, b_q = synth.normal("kb/wq", (288, 256), std=0.05).to(dev), synth.normal("kb/bq", (288,)).to(dev)
        for nm, w_, b_, cb in (("value", w_v, b_v, 16), ("value32", w_v, b_v, 32), ("qproj", w_q, b_q, 36)):
            res[f"linear_{nm}_standard"] = dict(ms=timeit(lambda: ops.linear_fused(x_tok, w_, b_)) * 1e3)
            res[f"linear_{nm}_blocked"] = dict(ms=timeit(lambda: ops.linear_blocked(x_tok, w_, b_, S, cb)) * 1e3)
    if not args.only or "mask" in args.only:
        Q, C, H, W = 100, 256, 184, 320
        e = synth.normal("kb/e", (T, Q, C)).to(dev)
        f = synth.normal("kb/f", (T, C, H, W)).to(dev)
        algm = 4.0 * (C * H * W + Q * C + Q * H * W) * T
        for impl, nm in ((1, "mask_decode_f32"), (2, "mask_decode_bf16x6")):
            ops.mask_decode_set_impl(impl)
            t = timeit(lambda: ops.mask_decode(e, f))
            res[nm] = dict(ms=t * 1e3, us_per_frame=t * 1e6 / T, GBps=algm / t / 1e9, frac_hbm=algm / t / HBM_PEAK,
                           TFLOPs=2.0 * Q * C * H * W * T / t / 1e12, impl=ops.mask_decode_last_impl())
            for (h, w) in shapes:
                fl = torch.nn.functional.interpolate(f, size=(h, w), mode="bilinear", align_corners=False).contiguous()
                t = timeit(lambda: ops.mask_decode_attn(e, fl))
                res[f"{nm}_attn_{h}x{w}"] = dict(ms=t * 1e3, impl=ops.mask_decode_last_impl())
        ops.mask_decode_set_impl(0)
        lows = {(h, w): torch.nn.functional.interpolate(f, size=(h, w), mode="bilinear", align_corners=False).contiguous()
                for (h, w) in shapes}
        for (h, w) in shapes:       # what the model runs (dispatch by size), and the chunked form of the exact-f32 kernel beside it
            fl = lows[(h, w)]
            t = timeit(lambda: ops.mask_decode_attn(e, fl))
            res[f"mask_decode_attn_default_{h}x{w}"] = dict(us=t * 1e6, impl=ops.mask_decode_last_impl())
            with ops.configured(mask_decode_impl=1, mask_decode_chunked=1):
                t = timeit(lambda: ops.mask_decode_attn(e, fl))
                res[f"mask_decode_attn_f32_chunked_{h}x{w}"] = dict(us=t * 1e6, impl=ops.mask_decode_last_impl())
            with ops.configured(mask_decode_impl=1):
                t = timeit(lambda: ops.mask_decode_attn(e, fl))
                res[f"mask_decode_attn_f32_oneshot_{h}x{w}"] = dict(us=t * 1e6, impl=ops.mask_decode_last_impl())
            for wt in (2, 3):
                with ops.configured(mask_decode_impl=2, mask_decode_wave_tiles=wt):
                    t = timeit(lambda: ops.mask_decode_attn(e, fl))
                    res[f"mask_decode_attn_bf16x6_wave_tiles{wt}_{h}x{w}"] = dict(us=t * 1e6, impl=ops.mask_decode_last_impl())
        e110 = synth.normal("kb/e110", (T, 110, C)).to(dev)
        t = timeit(lambda: ops.mask_decode(e110, f))
        res["mask_decode_bf16x6_q110"] = dict(ms=t * 1e3, impl=ops.mask_decode_last_impl())
        ops.mask_decode_set_impl(0)
        t = timeit(lambda: torch.einsum("tqc,tchw->tqhw", e, f))
        res["mask_decode_torch_einsum"] = dict(ms=t * 1e3, TFLOPs=2.0 * Q * C * H * W * T / t / 1e12)
    if not args.only or "linear" in args.only:
        # the encoder's token projections: 96 600 rows x 256 -> 256 / 288 (/ 1024 with ReLU)
        Mr = T * 19320
        xs = synth.normal("kb/lin/x", (Mr, 256)).to(dev)
        for N_, relu in ((256, False), (288, False), (1024, True)):
            w_ = synth.normal(f"kb/lin/w{N_}", (N_, 256), std=1 / 16).to(dev)
            b_ = synth.normal(f"kb/lin/b{N_}", (N_,)).to(dev)
            t = timeit(lambda: ops.linear_split(xs, w_, b_, relu=relu))
            fl = 2.0 * Mr * 256 * N_
            res[f"linear_split_{N_}"] = dict(ms=t * 1e3, TFLOPs=fl / t / 1e12, GBps=(Mr * (256 + N_) * 4.0) / t / 1e9)
            with ops.configured(linear_terms=3):
                t = timeit(lambda: ops.linear_split(xs, w_, b_, relu=relu))
            res[f"linear_f16x3_{N_}"] = dict(ms=t * 1e3, TFLOPs=fl / t / 1e12, GBps=(Mr * (256 + N_) * 4.0) / t / 1e9)
            with ops.configured(linear_terms=3, linear_ablate=1):
                t = timeit(lambda: ops.linear_split(xs, w_, b_, relu=relu))
            res[f"linear_f16x3_{N_}_no_row_pass"] = dict(ms=t * 1e3)
            if relu:
                t = timeit(lambda: torch._addmm_activation(b_, xs, w_.t(), use_gelu=False))
            else:
                t = timeit(lambda: torch.nn.functional.linear(xs, w_, b_))
            res[f"linear_aten_{N_}"] = dict(ms=t * 1e3, TFLOPs=fl / t / 1e12)
    if args.only and "swinlin" in args.only:
        # the Swin-T linears at 720p x 5 frames through the library GEMM: time against the HBM floor and the fp32 MFMA peak
        for stage, (tok, C) in enumerate(((184 * 320, 96), (92 * 160, 192), (46 * 80, 384), (23 * 40, 768)), 1):
            Mr = T * tok
            for nm, K_, N_ in (("qkv", C, 3 * C), ("proj", C, C), ("fc1", C, 4 * C), ("fc2", 4 * C, C)):
                xs = synth.normal(f"kb/sw/x{K_}/{Mr}", (Mr, K_)).to(dev)
                w_ = synth.normal(f"kb/sw/w{K_}x{N_}", (N_, K_), std=1 / 16).to(dev)
                b_ = synth.normal(f"kb/sw/b{N_}", (N_,)).to(dev)
                t = timeit(lambda: torch.nn.functional.linear(xs, w_, b_))
                byts = 4.0 * (Mr * K_ + Mr * N_ + N_ * K_)
                fl = 2.0 * Mr * K_ * N_
                res[f"swin_s{stage}_{nm}_{K_}x{N_}"] = dict(us=t * 1e6, TFLOPs=fl / t / 1e12, GBps=byts / t / 1e9,
                                                           hbm_floor_us=byts / HBM_PEAK * 1e6, frac_hbm=byts / t / HBM_PEAK)
                act = "gelu" if nm == "fc1" else None
                rs = torch.zeros(Mr, N_, device=dev) if nm == "fc2" else None
                y = ops.linear_fused(xs, w_, b_, act=act, residual=rs)
                if y is not None:
                    t2 = timeit(lambda: ops.linear_fused(xs, w_, b_, act=act, residual=rs))
                    res[f"swin_s{stage}_{nm}_{K_}x{N_}"]["fused_us"] = t2 * 1e6
                    from univs_amd.switches import override as _ov
                    with _ov(presplit_kmin=0), ops.configured(linear_terms=6):
                        if ops.linear_fused(xs, w_, b_, act=act, residual=rs) is not None:
                            t2 = timeit(lambda: ops.linear_fused(xs, w_, b_, act=act, residual=rs))
                            res[f"swin_s{stage}_{nm}_{K_}x{N_}"]["bf16x6_us"] = t2 * 1e6
                    with _ov(presplit_kmin=96):
                        t2 = timeit(lambda: ops.linear_fused(xs, w_, b_, act=act, residual=rs))
                    res[f"swin_s{stage}_{nm}_{K_}x{N_}"]["stream_us"] = t2 * 1e6
                    with _ov(presplit_kmin=0):
                        if ops.linear_fused(xs, w_, b_, act=act, residual=rs) is not None:
                            t2 = timeit(lambda: ops.linear_fused(xs, w_, b_, act=act, residual=rs))
                            res[f"swin_s{stage}_{nm}_{K_}x{N_}"]["resident_us"] = t2 * 1e6
                    if act:
                        t3 = timeit(lambda: torch.nn.functional.gelu(torch.nn.functional.linear(xs, w_, b_)))
                        res[f"swin_s{stage}_{nm}_{K_}x{N_}"]["library_plus_gelu_us"] = t3 * 1e6
                    if rs is not None:
                        t3 = timeit(lambda: rs + torch.nn.functional.linear(xs, w_, b_))
                        res[f"swin_s{stage}_{nm}_{K_}x{N_}"]["library_plus_add_us"] = t3 * 1e6
    if args.only and "mlp" in args.only:
        # two-Linear MLPs in one kernel (csrc/mlp_f16x3.hip) against the two fused Linears they replace: the encoder FFN and the
        # Swin Mlp + shortcut of the stages with C <= 256, at 720p x T frames
        for nm, Mr, C, Hd, act, with_res in (("encoder_ffn", T * 19320, 256, 1024, "relu", False), ("swin_s1_mlp", T * 184 * 320, 96, 384, "gelu", True),
                                           ("swin_s2_mlp", T * 92 * 160, 192, 768, "gelu", True), ("swin_s3_mlp", T * 46 * 80, 384, 1536, "gelu", True)):
            xs = synth.normal(f"kb/mlp/x{C}/{Mr}", (Mr, C)).to(dev)
            w1 = synth.normal(f"kb/mlp/w1/{C}", (Hd, C), std=C ** -0.5).to(dev)
            b1 = synth.normal(f"kb/mlp/b1/{C}", (Hd,), std=0.5).to(dev)
            w2 = synth.normal(f"kb/mlp/w2/{C}", (C, Hd), std=Hd ** -0.5).to(dev)
            b2 = synth.normal(f"kb/mlp/b2/{C}", (C,), std=0.5).to(dev)
            rs = synth.normal(f"kb/mlp/r{C}/{Mr}", (Mr, C)).to(dev) if with_res else None
            t2 = timeit(lambda: ops.linear_fused(ops.linear_fused(xs, w1, b1, act=act), w2, b2, residual=rs))
            t1 = timeit(lambda: ops.mlp_fused(xs, w1, b1, w2, b2, act, residual=rs))
            fl = 4.0 * Mr * C * Hd
            byts = 4.0 * Mr * C * (3 if with_res else 2)
            res[nm] = dict(fused_us=t1 * 1e6, two_kernels_us=t2 * 1e6, TFLOPs_fp32_equiv=fl / t1 / 1e12, f16_mfma_frac=3 * fl / t1 / 2.5e15,
                           hbm_frac=byts / t1 / HBM_PEAK)
            if C in (96, 256):       # timing experiments (results wrong): what the kernel costs without one of its parts
                for tag, abl in (("no_mfma_us", 2), ("no_lds_reads_us", 3), ("no_activation_split_us", 4)):
                    with ops.configured(linear_ablate=abl):
                        res[nm][tag] = timeit(lambda: ops.mlp_fused(xs, w1, b1, w2, b2, act, residual=rs)) * 1e6
    if args.only and "smallm" in args.only:
        # the Swin stage-3 / stage-4 Linears (few rows, wide): time by output features per pass and by workgroups along the rows
        from univs_amd.switches import override as _ov
        for nm, Mr, K_, N_, act, res_ in (("s3_qkv", T * 3680, 384, 1152, None, False), ("s3_proj", T * 3680, 384, 384, None, True),
                                          ("s3_fc1", T * 3680, 384, 1536, "gelu", False), ("s3_fc2", T * 3680, 1536, 384, None, True),
                                          ("s4_qkv", T * 920, 768, 2304, None, False), ("s4_fc1", T * 920, 768, 3072, "gelu", False),
                                          ("s4_fc2", T * 920, 3072, 768, None, True)):
            xs = synth.normal(f"kb/sm/x{K_}/{Mr}", (Mr, K_)).to(dev)
            w_ = synth.normal(f"kb/sm/w{K_}x{N_}", (N_, K_), std=K_ ** -0.5).to(dev)
            b_ = synth.normal(f"kb/sm/b{N_}", (N_,)).to(dev)
            rs = synth.normal(f"kb/sm/r{N_}/{Mr}", (Mr, N_)).to(dev) if res_ else None
            row = {}
            for kind, kmin in (("resident", 0), ("stream", 96)):
                if kind == "resident" and K_ > 768:
                    continue
                with _ov(presplit_kmin=kmin):
                    for rpp in (0, 64, 32):
                        for gx in (0, 16, 32, 64, 128):
                            with ops.configured(linear_rows_per_pass=rpp, linear_grid_x=gx):
                                if ops.linear_fused(xs, w_, b_, act=act, residual=rs) is None:
                                    continue
                                row[f"{kind}_r{rpp}_g{gx}"] = round(timeit(lambda: ops.linear_fused(xs, w_, b_, act=act, residual=rs), iters=10, warmup=3) * 1e6, 1)
            best = min(row, key=row.get)
            res[nm] = dict(best=best, best_us=row[best], default_resident=row.get("resident_r0_g0"), default_stream=row.get("stream_r0_g0"), all=row)
    if args.only and "conv" in args.only:
        from univs_amd.switches import override as _ov
        xc = synth.normal("kb/conv/x", (T, 256, 184, 320)).to(dev)
        wc = synth.normal("kb/conv/w", (256, 256, 3, 3), std=1 / 48).to(dev)
        fl = 2.0 * T * 184 * 320 * 256 * 2304
        t = timeit(lambda: ops.conv3x3(xc, wc))
        res["conv3x3_f16x3"] = dict(ms=t * 1e3, TFLOPs=fl / t / 1e12)
        t = timeit(lambda: torch.nn.functional.conv2d(xc, wc, None, 1, 1))
        res["conv3x3_miopen"] = dict(ms=t * 1e3, TFLOPs=fl / t / 1e12)
        xs = synth.normal("kb/ffn2/x", (T * 19320, 1024)).to(dev)
        w_ = synth.normal("kb/ffn2/w", (256, 1024), std=1 / 32).to(dev)
        b_ = synth.normal("kb/ffn2/b", (256,)).to(dev)
        t = timeit(lambda: ops.linear_fused(xs, w_, b_))
        res["ffn2_1024x256_f16x3"] = dict(us=t * 1e6, TFLOPs=2.0 * T * 19320 * 1024 * 256 / t / 1e12)
        t = timeit(lambda: torch.nn.functional.linear(xs, w_, b_))
        res["ffn2_1024x256_aten"] = dict(us=t * 1e6, TFLOPs=2.0 * T * 19320 * 1024 * 256 / t / 1e12)
    if args.only and "bigk" in args.only:
        for nm, Mr, K_, N_ in (("enc_ffn2", T * 19320, 1024, 256), ("swin_s3_fc2", T * 3680, 1536, 384), ("swin_s4_fc2", T * 920, 3072, 768)):
            xs = synth.normal(f"kb/bk/x{K_}/{Mr}", (Mr, K_)).to(dev)
            w_ = synth.normal(f"kb/bk/w{K_}x{N_}", (N_, K_), std=1 / 16).to(dev)
            b_ = synth.normal(f"kb/bk/b{N_}", (N_,)).to(dev)
            t = timeit(lambda: torch.nn.functional.linear(xs, w_, b_))
            res[nm] = dict(lib_us=t * 1e6)
            y = ops.linear_fused(xs, w_, b_)
            if y is not None:
                res[nm]["fused_us"] = timeit(lambda: ops.linear_fused(xs, w_, b_)) * 1e6
    if not args.only or "win" in args.only:
        # Swin-T stage 1 at 720p: 27x46 windows of 49 tokens, 3 heads, per frame
        nW, nH, ntok, hd = 27 * 46, 3, 49, 32
        qkv = synth.normal("kb/qkv", (T * nW, ntok, 3, nH, hd)).to(dev)
        bias = synth.normal("kb/bias", (nH, ntok, ntok)).to(dev)
        mask = torch.zeros(nW, ntok, ntok, device=dev)
        t = timeit(lambda: ops.window_attention(qkv, bias, mask, nW, hd ** -0.5))
        byts = (qkv.numel() + qkv.numel() / 3) * 4.0
        res["window_attn_stage1"] = dict(ms=t * 1e3, GBps=byts / t / 1e9, frac_hbm=byts / t / HBM_PEAK)
        # image mode (what the backbone calls), the four Swin-T stages at 720p, plain and shifted windows
        for (H_, W_, nh) in ((184, 320, 3), (92, 160, 6), (46, 80, 12), (23, 40, 24)):
            ws = 7
            q = synth.normal(f"kb/qkvi/{H_}", (T, H_ * W_, 3, nh, hd)).to(dev)
            qb = synth.normal(f"kb/qb/{nh}", (3 * nh * hd,)).to(dev)
            bi = synth.normal(f"kb/biasi/{nh}", (nh, ntok, ntok)).to(dev)
            nWi = ((H_ + ws - 1) // ws) * ((W_ + ws - 1) // ws)
            mk = torch.zeros(nWi, ntok, ntok, device=dev)
            byts = (q.numel() + q.numel() / 3) * 4.0
            for shift in (0, 3):
                for v1 in ("0", "1"):
                    with ops.configured(window_attn_v1=int(v1)):
                        t = timeit(lambda: ops.window_attention_image(q, qb, bi, mk if shift else None, H_, W_, ws, shift, hd ** -0.5))
                    res[f"window_attn_image_{H_}x{W_}_shift{shift}" + ("_v1" if v1 == "1" else "")] = dict(
                        ms=t * 1e3, us_per_frame=t * 1e6 / T, GBps=byts / t / 1e9, frac_hbm=byts / t / HBM_PEAK)
                t = timeit(lambda: ops.window_attention_image(q, qb, bi, mk if shift else None, H_, W_, ws, shift, hd ** -0.5, mma="f16x3"))
                res[f"window_attn_image_{H_}x{W_}_shift{shift}_f16x3"] = dict(
                    ms=t * 1e3, us_per_frame=t * 1e6 / T, GBps=byts / t / 1e9, frac_hbm=byts / t / HBM_PEAK)
    if not args.only or "win12" in args.only:
        # config 5: Swin-L at 1080p (1088 x 1920 padded), 12 x 12 windows, the four stages, 2 frames; exact-f32 vs fp16 operands
        hd, ws, Tw = 32, 12, 2
        for (H_, W_, nh) in ((272, 480, 6), (136, 240, 12), (68, 120, 24), (34, 60, 48)):
            q = synth.normal(f"kb/qkv12/{H_}", (Tw, H_ * W_, 3, nh, hd)).to(dev)
            qb = synth.normal(f"kb/qb12/{nh}", (3 * nh * hd,)).to(dev)
            bi = synth.normal(f"kb/bias12/{nh}", (nh, ws * ws, ws * ws)).to(dev)
            nWi = ((H_ + ws - 1) // ws) * ((W_ + ws - 1) // ws)
            mk = torch.zeros(nWi, ws * ws, ws * ws, device=dev)
            byts = (q.numel() + q.numel() / 3) * 4.0
            for shift in (0, 6):
                for mma in ("f32", "f16"):
                    t = timeit(lambda: ops.window_attention_image(q, qb, bi, mk if shift else None, H_, W_, ws, shift, hd ** -0.5, mma=mma))
                    res[f"window12_{H_}x{W_}_shift{shift}_{mma}"] = dict(
                        ms=t * 1e3, us_per_frame=t * 1e6 / Tw, GBps=byts / t / 1e9, frac_hbm=byts / t / HBM_PEAK)
    if not args.only or "resample" in args.only:
        f = synth.normal("kb/f", (T, 256, 184, 320)).to(dev)
        for (h, w) in ((92, 160), (46, 80), (23, 40)):
            t = timeit(lambda: ops.bilinear_resample(f, (h, w)))
            byts = (min(f.numel(), 4 * T * 256 * h * w) + T * 256 * h * w) * 4.0     # taps actually read + output
            res[f"resample_{h}x{w}"] = dict(ms=t * 1e3, GBps=byts / t / 1e9, frac_hbm=byts / t / HBM_PEAK)
            t = timeit(lambda: torch.nn.functional.interpolate(f, size=(h, w), mode="bilinear", align_corners=False))
            res[f"resample_{h}x{w}_aten"] = dict(ms=t * 1e3)
    def fmt(vv):
        if isinstance(vv, (int, float)) and not isinstance(vv, bool):
            return round(vv, 4) if abs(vv) > 1e-3 or vv == 0 else float(f'{vv:.3e}')
        return vv
    for k, v in res.items():
        print(k, json.dumps({kk: fmt(vv) for kk, vv in v.items()}))


if __name__ == "__main__":
    main()
