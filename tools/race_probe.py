"""Run-to-run bit identity of every stage of the config-2 first clip WHILE other processes load the same GPU (GPU box):
    for i in 1 2; do python tools/race_probe.py --tag p$i & done; wait
A stage whose output differs from its own first run under contention has a race (or reads something uninitialised)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univs_amd import workloads as cases  # noqa: E402


def flat(o):
    if isinstance(o, torch.Tensor):
        return [o]
    if isinstance(o, dict):
        return [t for k in sorted(o) for t in flat(o[k])]
    if isinstance(o, (list, tuple)):
        return [t for x in o for t in flat(x)]
    return []


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="p")
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--no-swin", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    swin, head = cases.build_model(dev)
    x = cases.preprocess(cases.cfg2_frames()).to(dev)
    tg = cases.targets_first_clip(cases.CFG2)
    tg = [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in tg[0].items()}]
    with torch.no_grad():
        feats0 = swin(x)
        pd0 = head.pixel_decoder.forward_features(feats0)
        out0 = head(feats0, targets=[dict(tg[0])])
        torch.cuda.synchronize()
        names_pd = ["mask_features", "fpn_out(1/4)", "enc_out[0](1/32)", "ms0", "ms1", "ms2"]
        bad_pd = [0] * 6
        bad = {"swin": 0, "pixel_decoder": 0, "head": 0}
        worst = {"swin": 0.0, "pixel_decoder": 0.0, "head": 0.0}
        t0 = time.time()
        for it in range(args.iters):
            f = swin(x) if not args.no_swin else feats0
            pd = head.pixel_decoder.forward_features(feats0)
            for k_, (p_, q_) in enumerate(zip(flat(pd), flat(pd0))):
                if not torch.equal(p_, q_):
                    bad_pd[k_] += 1
            out = head(feats0, targets=[dict(tg[0])])
            for name, a, b in (("swin", f, feats0), ("pixel_decoder", pd, pd0), ("head", {k: out[k] for k in ("pred_masks", "pred_logits", "pred_embds")}, {k: out0[k] for k in ("pred_masks", "pred_logits", "pred_embds")})):
                fa, fb = flat(a), flat(b)
                eq = all(torch.equal(p, q) for p, q in zip(fa, fb))
                if not eq:
                    bad[name] += 1
                    worst[name] = max(worst[name], max((p.float() - q.float()).abs().max().item() for p, q in zip(fa, fb)))
        torch.cuda.synchronize()
    print(f"{args.tag}: pixel-decoder outputs that differed: {dict(zip(names_pd, bad_pd))}")
    print(f"{args.tag}: {args.iters} iterations in {time.time() - t0:.1f} s; runs that differ from the first: {bad}; largest difference {worst}", flush=True)


if __name__ == "__main__":
    main()
