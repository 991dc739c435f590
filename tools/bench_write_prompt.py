"""`write_prompt_predictions_into_annotations_per_clip` alone at config 2's size with N tracked entities (the bench's synthetic video ends with
one entity; real videos track several): time per call with a device synchronisation either side, and the share of it that is device time.
    python tools/bench_write_prompt.py [--entities 10]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univs_amd import synth  # noqa: E402
from univs_amd import workloads as cases  # noqa: E402
from univs_amd.inference.video_entity import InferenceVideoEntity  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--entities", type=int, default=10)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--kernels", action="store_true", help="torch.profiler table of one call's device kernels")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    case = cases.CFG2
    T, N, K, C, hist = case["T"], args.entities, 40, 256, 10
    H, W, h, w = 736, 1280, 184, 320
    loop = InferenceVideoEntity(
        hidden_dim=256, num_queries=case["Q"], overlap_threshold_entity=0.5, stability_score_thresh=0.5, size_divisibility=32,
        pixel_mean=synth.PIXEL_MEAN, pixel_std=synth.PIXEL_STD, num_frames=T, test_topk_per_image=100, apply_cls_thres=0.25,
        box_nms_thresh=0.85, num_frames_window_test=20, clip_stride=1, num_prev_frames_memory=5,
        video_unified_inference_entities="", temporal_consistency_threshold=0.25, detect_newly_object_threshold=0.1,
        detect_newly_interval_frames=1, custom_videos_enable=False).to(dev)
    g = torch.Generator().manual_seed(0)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    low = torch.stack([torch.stack([6.0 * (1.0 - (((yy - 30 - 12 * n) ** 2 + (xx - 40 - 25 * n - 2 * t) ** 2) / (18 + n) ** 2)) for t in range(T)]) for n in range(N)])
    embds = torch.nn.functional.normalize(torch.randn(N, 1, C, generator=g), dim=-1)

    def fresh():
        ml = torch.zeros(N, hist, H, W, device=dev)
        tv = {"sub_task": "vis", "embds": embds.repeat(1, hist, 1).to(dev), "logits": torch.rand(N, hist, K, generator=g).to(dev),
              "mask_logits": ml, "masks": ml.gt(0).float(), "occurrence": torch.zeros(N, hist, device=dev), "boxes": torch.zeros(N, hist, 4, device=dev),
              "mask_quality_scores": torch.zeros(N, device=dev)}
        out = {"pred_masks": low.to(dev).clone(), "pred_embds": (embds.repeat(1, T, 1) + 0.01 * torch.randn(N, T, C, generator=g)).to(dev),
               "pred_logits": torch.rand(N, K, generator=g).to(dev)}
        return out, [tv]
    with torch.no_grad():
        for _ in range(3):
            out, tg = fresh()
            loop.write_prompt_predictions_into_annotations_per_clip(5, out, tg, (H, W), (720, 1280), 1)
        torch.cuda.synchronize()
        wall = dev_ms = 0.0
        kept = None
        out0, tg = fresh()
        state0 = {k: v.clone() for k, v in tg[0].items() if isinstance(v, torch.Tensor)}
        for _ in range(args.reps):
            # the same tensors every time, reset in place: fresh allocations of ~1 GB per call put the caching allocator's hipMalloc /
            # hipFree into the timed region (8.9 ms per call measured that way against 1.8 ms of kernels)
            out = {k: v.clone() for k, v in out0.items()}
            for k, v in state0.items():
                if tg[0][k].shape == v.shape and tg[0][k].dtype == v.dtype:
                    tg[0][k].copy_(v)
                else:
                    tg[0][k] = v.clone()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            loop.write_prompt_predictions_into_annotations_per_clip(5, out, tg, (H, W), (720, 1280), 1)
            e1.record()
            torch.cuda.synchronize()
            wall += time.perf_counter() - t0
            dev_ms += e0.elapsed_time(e1)
            kept = int((tg[0]["occurrence"][:, -T:].sum(1) > 0).sum())
    if args.kernels:
        from torch.profiler import ProfilerActivity, profile
        with torch.no_grad():
            out, tg = fresh()
            torch.cuda.synchronize()
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                loop.write_prompt_predictions_into_annotations_per_clip(5, out, tg, (H, W), (720, 1280), 1)
                torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=14, max_name_column_width=70))
    print(f"{N} entities at {H} x {W}, T = {T}: {wall / args.reps * 1e3:.2f} ms per call (events around the call: {dev_ms / args.reps:.2f} ms); entities kept {kept} of {N}")


if __name__ == "__main__":
    main()
