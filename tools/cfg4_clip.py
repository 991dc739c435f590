"""Steady-state clips of BASELINE config 4 for profiling (Swin-B, T=5 @ 720p, 200 queries + 4 referring expressions: lang->vision
cross-attention, ProCA, 'sep-blocked' self-attention), backbone + head each, as bench.py's `config4_swinb_refvos` leg runs them.
    rocprofv3 --kernel-trace --output-format csv -d out -- python tools/cfg4_clip.py --clips 8
    python tools/clip_breakdown.py out/*/*_kernel_trace.csv --skip 3 --last 4"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from univs_amd import ops, runtime, synth  # noqa: E402
from univs_amd import workloads as cases  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=8)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    runtime.enable_tuned_gemms()
    c4 = cases.CFG4
    swin = cases.build_swin(dev, variant=cases.SWIN_B)
    head = cases.build_head(c4, dev, return_aux=False, **cases.CFG4_DECODER)
    frames = cases.cfg2_frames().to(dev)
    mean = torch.tensor(synth.PIXEL_MEAN, device=dev)
    std = torch.tensor(synth.PIXEL_STD, device=dev)
    tg = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in cases.cfg4_targets(c4)[0].items()}
    with torch.no_grad():
        def step():
            return head(swin(ops.normalize_pad(frames, mean, std, pad_to=(736, 1280))), targets=[dict(tg)])
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.clips):
            out = step()
        torch.cuda.synchronize()
    print(f"config 4: {(time.perf_counter() - t0) / args.clips * 1e3:.2f} ms per clip, queries {out['pred_masks'].shape[1]}")


if __name__ == "__main__":
    main()
