"""Where inside the pixel decoder's encoder the run-to-run difference under GPU contention starts (see tools/race_probe.py):
    for i in 1 2; do python tools/race_probe3.py --tag p$i & done; wait"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univs_amd import ops  # noqa: E402
from univs_amd import workloads as cases  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="p")
    ap.add_argument("--iters", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    swin, head = cases.build_model(dev)
    pdm = head.pixel_decoder
    x = cases.preprocess(cases.cfg2_frames()).to(dev)
    cap = {}
    orig_tok = ops.tokens_from_nchw

    def tok(*a, **k):
        r = orig_tok(*a, **k)
        if r is not None:
            cap["tokens"], cap["query0"] = r[0].clone(), r[1].clone()
        return r
    ops.tokens_from_nchw = tok
    enc = pdm.transformer.encoder
    for li, layer in enumerate(enc.layers):
        layer.register_forward_hook(lambda m, i, o, li=li: cap.__setitem__(f"layer{li}", (o[0] if isinstance(o, tuple) else o).clone()))
        layer.self_attn.register_forward_hook(lambda m, i, o, li=li: cap.__setitem__(f"layer{li}.attn", (o[0] if isinstance(o, tuple) else o).clone()))

    def run(feats):
        cap.clear()
        pdm.forward_features(feats)
        return dict(cap)
    with torch.no_grad():
        feats0 = swin(x)
        ref = run(feats0)
        torch.cuda.synchronize()
        bad = {k: 0 for k in ref}
        first_bad = {}
        for it in range(args.iters):
            swin(x)
            got = run(feats0)
            order = ["tokens", "query0"] + [n for li in range(len(enc.layers)) for n in (f"layer{li}.attn", f"layer{li}")]
            seen = False
            for k in order:
                if k in ref and not torch.equal(got[k], ref[k]):
                    bad[k] += 1
                    if not seen:
                        first_bad[k] = first_bad.get(k, 0) + 1
                        seen = True
        torch.cuda.synchronize()
    print(f"{args.tag}: differed from the first run (of {args.iters}): {bad}\n{args.tag}: FIRST step that differed, by iteration count: {first_bad}", flush=True)


if __name__ == "__main__":
    main()
