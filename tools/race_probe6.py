"""Which kernel disturbs which under GPU sharing?  A VICTIM process runs small transposes (with / without the row affine and the
addend) and an ATen formulation of the same thing, and compares every run with its first; an AGGRESSOR process loops ONE kind of
kernel on the same GPU meanwhile.
    python tools/race_probe6.py --role aggressor --kind mlp96 --flag /tmp/f &     python tools/race_probe6.py --role victim --flag /tmp/f
`--role both`: one process, the aggressor on a second stream."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univs_amd import ops  # noqa: E402

KINDS = ["none", "matmul", "aten_ew", "transpose", "ln", "linear", "mlp96", "mlp384", "wattn", "swin", "pixdec"]


def make_aggressor(kind, dev):
    g = torch.Generator(device="cpu").manual_seed(7)

    def rn(*s, scale=1.0):
        return (torch.randn(*s, generator=g) * scale).to(dev)
    if kind == "none":
        return lambda: None
    if kind == "matmul":
        a = rn(4096, 4096)
        return lambda: a @ a
    if kind == "aten_ew":
        a = rn(5, 256, 14720)
        return lambda: a + 1.0
    if kind == "transpose":
        a = rn(5, 256, 14720)
        return lambda: ops.transpose_last2(a)
    M = 5 * 184 * 320
    if kind == "ln":
        x, w, b = rn(M, 96), rn(96), rn(96)
        return lambda: ops.layer_norm(x, w, b)
    if kind == "linear":
        x, w, b = rn(M, 96), rn(288, 96, scale=0.1), rn(288)
        return lambda: ops.linear_fused(x, w, b)
    if kind in ("mlp96", "mlp384"):
        C, m = (96, M) if kind == "mlp96" else (384, 5 * 46 * 80)
        x, w1, b1, w2, b2 = rn(m, C), rn(4 * C, C, scale=0.1), rn(4 * C), rn(C, 4 * C, scale=0.05), rn(C)
        lw, lb = rn(C), rn(C)
        return lambda: ops.mlp_fused(x, w1, b1, w2, b2, "gelu", residual=x, ln=(lw, lb, 1e-5))
    if kind == "wattn":
        qkv, qb, bias = rn(5, 184 * 320, 3, 3, 32), rn(288), rn(3, 49, 49)
        return lambda: ops.window_attention_image(qkv, qb, bias, None, 184, 320, 7, 0, 32 ** -0.5, mma="f16x3")
    from univs_amd import workloads as cases
    swin, head = cases.build_model(dev)
    x = cases.preprocess(cases.cfg2_frames()).to(dev)
    if kind == "swin":
        return lambda: swin(x)
    if kind == "pixdec":
        with torch.no_grad():
            feats = swin(x)
        return lambda: head.pixel_decoder.forward_features(feats)
    raise ValueError(kind)


def make_victims(dev):
    g = torch.Generator(device="cpu").manual_seed(3)
    T, C = 5, 256
    hws = [920, 3680, 14720]
    xs = [torch.randn(T, C, hw // 40, 40, generator=g).to(dev) for hw in hws]
    affs = [torch.stack([torch.rand(T * C, generator=g) + 0.5, torch.randn(T * C, generator=g)], 1).contiguous().to(dev) for _ in hws]
    pos = torch.randn(1, sum(hws), C, generator=g).to(dev)

    def aten():
        ys = [(x.flatten(2) * a[:, 0].view(T, C, 1) + a[:, 1].view(T, C, 1)).transpose(1, 2) for x, a in zip(xs, affs)]
        src = torch.cat(ys, 1)
        return src, src + pos
    return {
        "affine + addend (our)": lambda: ops.tokens_from_nchw(xs, affs, pos),
        "affine only (our)": lambda: ops.tokens_from_nchw(xs, affs, None)[:1],
        "addend only (our)": lambda: ops.tokens_from_nchw(xs, [None] * 3, pos),
        "plain transposes (our)": lambda: ops.tokens_from_nchw(xs, [None] * 3, None)[:1],
        "affine + addend (ATen)": aten,
    }, xs, affs


def describe(name, got, ref, xs, affs):
    d = (got != ref).nonzero()
    ch, tok = d[:, 2], d[:, 1]
    out = f"   {name}: {len(d)} elements; channel % 4 histogram {torch.bincount(ch % 4, minlength=4).tolist()}; token % 4 {torch.bincount(tok % 4, minlength=4).tolist()}; " \
          f"frames {sorted(set(d[:, 0].tolist()))}; tokens {int(tok.min())}..{int(tok.max())}"
    starts = [0, 920, 4600]
    for t, s, c in d[:4].tolist():
        lv = 2 if s >= 4600 else (1 if s >= 920 else 0)
        x = xs[lv].flatten(2)[t, c, s - starts[lv]].item()
        sc, bi = affs[lv][t * 256 + c].tolist()
        out += f"\n      (t {t}, token {s}, c {c}): ref {ref[t, s, c].item():.6f} got {got[t, s, c].item():.6f}; x {x:.6f} scale {sc:.6f} bias {bi:.6f}; x * scale + bias {x * sc + bi:.6f}"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--role", choices=["victim", "aggressor", "both"], required=True)
    ap.add_argument("--kind", default="none", choices=KINDS)
    ap.add_argument("--flag", default="/tmp/race6_flag")
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--seconds", type=float, default=120.0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    if args.role == "aggressor":
        f = make_aggressor(args.kind, dev)
        f()
        torch.cuda.synchronize()
        open(args.flag + ".ready", "w").close()
        t0 = time.time()
        n = 0
        while not os.path.exists(args.flag + ".done") and time.time() - t0 < args.seconds:
            for _ in range(8):
                f()
            torch.cuda.synchronize()
            n += 8
        print(f"aggressor {args.kind}: {n} calls in {time.time() - t0:.1f} s", flush=True)
        return
    victims, xs, affs = make_victims(dev)
    refs = {k: [t.clone() for t in v()] for k, v in victims.items()}
    torch.cuda.synchronize()
    side = None
    if args.role == "both":
        f = make_aggressor(args.kind, dev)
        side = torch.cuda.Stream()
    else:
        t0 = time.time()
        while not os.path.exists(args.flag + ".ready") and time.time() - t0 < 180:
            time.sleep(0.2)
    bad = {k: 0 for k in victims}
    shown = 0
    for it in range(args.iters):
        if side is not None:
            with torch.cuda.stream(side):
                for _ in range(4):
                    f()
        for k, v in victims.items():
            got = v()
            if not all(torch.equal(a, b) for a, b in zip(got, refs[k])):
                bad[k] += 1
                if shown < 3 and "our" in k:
                    shown += 1
                    print(describe(k, got[0], refs[k][0], xs, affs), flush=True)
    torch.cuda.synchronize()
    if args.role == "victim":
        open(args.flag + ".done", "w").close()
    print(f"victim against [{args.kind}] ({args.role}): runs of {args.iters} that differed from the first: {bad}", flush=True)


if __name__ == "__main__":
    main()
