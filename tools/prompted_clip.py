"""Steady-state clips for profiling: N first clips (100 learnable queries, the BASELINE config-2 clip) followed by N
PROMPTED clips (second clip of a video: 10 tracked entities -> prompt sampler, memory pool, ProCA, 110 queries), backbone +
head each, as bench.py's `steady_state_with_prompts` leg runs them.  Under `rocprofv3 --kernel-trace` the trace is cut into
clips by tools/clip_breakdown.py (anchor: the 6 MSDeformAttn launches of a clip): the LAST clips are the prompted ones.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03_prompted -- python tools/prompted_clip.py --clips 10
    python tools/clip_breakdown.py gpurun_out/r03_prompted/*/*_kernel_trace.csv --last 8              # prompted clips
    python tools/clip_breakdown.py gpurun_out/r03_prompted/*/*_kernel_trace.csv --skip 3 --last 6     # first clips
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from univs_amd import workloads as cases  # noqa: E402
from univs_amd import runtime  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=10)
    ap.add_argument("--entities", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    runtime.enable_tuned_gemms()
    swin, head = cases.build_model(dev)
    case = dict(cases.CFG2, H=736, W=1280)
    x = cases.preprocess(cases.cfg2_frames()).to(dev)
    first = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in cases.targets_first_clip(case)[0].items()}
    tv = cases.targets_with_entities(case, first_frame_idx=1, n_ent=args.entities)[0]
    prompted = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in tv.items()}
    with torch.no_grad():
        for name, tg in (("first clip", first), (f"prompted clip ({args.entities} entities)", prompted)):
            for _ in range(2):
                torch.manual_seed(0)
                tgl = [dict(tg)]
                head.prefetch_prompts(tgl, x.shape[0])
                head(swin(x), targets=tgl)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            enq = 0.0
            for _ in range(args.clips):
                torch.manual_seed(0)
                h0 = time.perf_counter()
                tgl = [dict(tg)]
                head.prefetch_prompts(tgl, x.shape[0])
                out = head(swin(x), targets=tgl)
                enq += time.perf_counter() - h0
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.clips
            print(f"{name}: {1e3 * dt:.2f} ms per clip (host enqueue {1e3 * enq / args.clips:.2f} ms), queries {out['pred_masks'].shape[1]}",
                  flush=True)


if __name__ == "__main__":
    main()
