"""Sanity checks under GPU contention (two processes): do plain ATen results change from run to run when OUR backbone runs in the same loop?
    for i in 1 2; do python tools/race_probe5.py --tag p$i & done; wait"""
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
    ap.add_argument("--filler", default="swin", choices=["swin", "matmul", "none"])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    swin, head = cases.build_model(dev)
    x = cases.preprocess(cases.cfg2_frames()).to(dev)
    a = torch.randn(5, 256, 14720, device=dev)
    big = torch.randn(4096, 4096, device=dev)
    with torch.no_grad():
        def work():
            return {"aten transpose copy": a.transpose(1, 2).contiguous(), "aten add": a + 1.0, "our transpose": ops.transpose_last2(a),
                    "aten sum": a.sum(-1)}
        ref = {k: v.clone() for k, v in work().items()}
        torch.cuda.synchronize()
        bad = {k: 0 for k in ref}
        bad_after_sync = {k: 0 for k in ref}
        for it in range(args.iters):
            if args.filler == "swin":
                swin(x)
            elif args.filler == "matmul":
                for _ in range(6):
                    big @ big
            got = work()
            for k in ref:
                if not torch.equal(got[k], ref[k]):
                    bad[k] += 1
            torch.cuda.synchronize()
            for k in ref:
                if not torch.equal(got[k], ref[k]):
                    bad_after_sync[k] += 1
        torch.cuda.synchronize()
    print(f"{args.tag} [{args.filler}]: differed from the first run: {bad}; the same tensors compared again after a device synchronisation: {bad_after_sync}", flush=True)


if __name__ == "__main__":
    main()
