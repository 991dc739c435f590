"""Which launch of the Swin backbone disturbs the affine transposes when it runs BESIDE them (one process, two streams)?  Every ops.* call
of one backbone forward is recorded (graphs off), de-duplicated by name + shapes, and replayed alone on a side stream while the victims
of race_probe6 run on the main stream.     python tools/race_probe7.py [--iters 30]"""
import argparse
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univs_amd import ops  # noqa: E402
from univs_amd import workloads as cases  # noqa: E402
from univs_amd.switches import SWITCHES  # noqa: E402
from race_probe6 import make_victims, describe  # noqa: E402


def sig(a):
    if isinstance(a, torch.Tensor):
        return ("T",) + tuple(a.shape)
    if isinstance(a, (list, tuple)):
        return tuple(sig(x) for x in a)
    if isinstance(a, (int, float, str, bool, type(None))):
        return a
    return type(a).__name__


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--what", default="swin", choices=["swin", "head"])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    swin, head = cases.build_model(dev)
    x = cases.preprocess(cases.cfg2_frames()).to(dev)
    SWITCHES.graphs = False
    calls = {}
    originals = {}
    for name, fn in list(vars(ops).items()):
        if isinstance(fn, types.FunctionType) and not name.startswith("_") and fn.__module__ == ops.__name__:
            def wrap(fn=fn, name=name):
                def rec(*a, **k):
                    key = (name, sig(a), tuple(sorted((kk, sig(v)) for kk, v in k.items())))
                    calls.setdefault(key, (fn, a, k))
                    return fn(*a, **k)
                return rec
            originals[name] = fn
            setattr(ops, name, wrap())
    if args.what == "swin":
        swin(x)
    else:
        feats = swin(x)
        calls.clear()
        head.pixel_decoder.forward_features(feats)
    for name, fn in originals.items():
        setattr(ops, name, fn)
    torch.cuda.synchronize()
    skip = {"needs_grad", "presplit_weights", "presplit_generation", "get_config", "configure", "configured", "invalidate_presplit"}
    cands = [(k, v) for k, v in calls.items() if k[0] not in skip]
    print(f"{len(cands)} distinct ops.* calls in one {args.what} forward", flush=True)
    victims, xs, affs = make_victims(dev)
    victims = {k: v for k, v in victims.items() if k in ("affine + addend (our)", "affine only (our)", "plain transposes (our)")}
    refs = {k: [t.clone() for t in v()] for k, v in victims.items()}
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    shown = 0
    whole = (("whole forward (eager)",), (lambda: swin(x), (), {}))
    for key, (fn, a, k) in [whole] + cands:
        bad = {n: 0 for n in victims}
        for it in range(args.iters):
            with torch.cuda.stream(side):
                for _ in range(4):
                    fn(*a, **k)
            for n, v in victims.items():
                got = v()
                if not all(torch.equal(p, q) for p, q in zip(got, refs[n])):
                    bad[n] += 1
                    if shown < 2:
                        shown += 1
                        print(describe(n, got[0], refs[n][0], xs, affs), flush=True)
        torch.cuda.synchronize()
        tag = "  <<<<" if any(bad.values()) else ""
        print(f"{str(key)[:230]}: {list(bad.values())}{tag}", flush=True)


if __name__ == "__main__":
    main()
