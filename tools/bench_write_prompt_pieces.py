"""Device time of the pieces of `write_prompt_predictions_into_annotations_per_clip` at N entities (events around each piece)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univs_amd import ops  # noqa: E402
from univs_amd.inference import video_entity as ve  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
T, hist, H, W = 5, 10, 736, 1280
low = torch.randn(N, T, 184, 320, device=dev) * 3
ml = torch.randn(N, hist, H, W, device=dev)
mk = ml.gt(0).float()
score = torch.rand(N, device=dev)


def timed(name, fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"  {e0.elapsed_time(e1) / reps:7.3f} ms  {name}")


with torch.no_grad():
    pm = ve._resize(low, (H, W))
    cur = pm[:, :, :720, :1280]
    recent = ml[:, -T:]
    idx = torch.arange(N, device=dev)
    print(f"{N} entities:")
    timed("_resize (bilinear x 4)", lambda: ve._resize(low, (H, W)))
    timed("mask_stats(pred_masks, valid)", lambda: ops.mask_stats(pm, valid=(720, 1280)))
    timed("mask_stats(pred_masks)", lambda: ops.mask_stats(pm))
    timed("mask_stats(recent view)", lambda: ops.mask_stats(recent))
    timed("recent += pred_masks", lambda: recent.add_(pm))
    timed("old: x[idx, -T:] += m (index, add, index_put)", lambda: ml.__setitem__((idx, slice(-T, None)), ml[idx, -T:] + pm[idx]))
    timed("refresh recent masks (slice)", lambda: mk[:, -T:].copy_(ml[:, -T:].gt(0.0)))
    timed("old: masks = logits.gt(0).float() (whole history)", lambda: ml.gt(0.0).float())

    def ownership():
        prob = cur.sigmoid().flatten(1)
        fg = prob > 0.5
        owner = (score.view(-1, 1) * prob).argmax(0)
        owner = torch.where((prob < 0.5).all(0), torch.full_like(owner, -1), owner)
        own = owner[None] == torch.arange(len(prob), device=prob.device).view(-1, 1)
        return own.sum(1) / fg.sum(1).clamp(min=1), (own & fg).sum(1)
    timed("ownership chain (sigmoid .. counts)", ownership)
    timed("  of it: cur.sigmoid().flatten(1)", lambda: cur.sigmoid().flatten(1))
    prob = cur.sigmoid().flatten(1)
    timed("  of it: (score * prob).argmax(0)", lambda: (score.view(-1, 1) * prob).argmax(0))
    timed("  of it: (prob < 0.5).all(0)", lambda: (prob < 0.5).all(0))
