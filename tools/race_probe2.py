"""Which step of the pixel decoder changes from run to run while another process loads the GPU and the backbone runs in the same loop
(tools/race_probe.py found: every output of the pixel decoder, only with the backbone in the loop):
    for i in 1 2; do python tools/race_probe2.py --tag p$i & done; wait"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univs_amd import ops  # noqa: E402
from univs_amd import workloads as cases  # noqa: E402
from univs_amd.switches import SWITCHES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="p")
    ap.add_argument("--iters", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    swin, head = cases.build_model(dev)
    pdm = head.pixel_decoder
    x = cases.preprocess(cases.cfg2_frames()).to(dev)

    def steps(features):
        out = {}
        srcs, pos, affines = [], [], []
        for idx, f in enumerate(pdm.transformer_in_features[::-1]):
            xf = features[f].float()
            conv, gn = pdm.input_proj[idx][0], pdm.input_proj[idx][1]
            raw = ops.conv1x1(xf, conv.weight, conv.bias)
            raw = conv(xf) if raw is None else raw
            srcs.append(raw)
            affines.append(ops.group_norm_affine(raw, gn.num_groups, gn.weight, gn.bias, gn.eps))
            pos.append(pdm._pos(xf))
            out[f"conv1x1[{idx}]"] = raw
            out[f"gn_affine[{idx}]"] = torch.cat([t.flatten() for t in affines[-1]]) if isinstance(affines[-1], (tuple, list)) else affines[-1]
        y, _, _ = pdm.transformer(srcs, pos, affines)
        out["encoder"] = y
        return out
    with torch.no_grad():
        feats0 = swin(x)
        ref = steps(feats0)
        torch.cuda.synchronize()
        bad = {k: 0 for k in ref}
        for it in range(args.iters):
            swin(x)
            got = steps(feats0)
            for k in ref:
                if not torch.equal(got[k], ref[k]):
                    bad[k] += 1
        torch.cuda.synchronize()
    print(f"{args.tag}: steps that differed from their first run in {args.iters} iterations: {bad}", flush=True)


if __name__ == "__main__":
    main()
