"""Where a VIDEO's time goes (GPU box): the sliding clip loop of inference/video_entity.py over a 20-frame 720p video, every stage of a
clip timed with a device synchronisation either side (so the figures add up to the un-overlapped cost), then a torch.profiler table of
the device kernels by total time.   python tools/prof_video_loop.py [--window 20] [--frames 20]"""
import argparse
import collections
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univs_amd import synth  # noqa: E402
from univs_amd import workloads as cases  # noqa: E402
from univs_amd.inference.video_entity import InferenceVideoEntity, normalized_image_list  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window", type=int, default=20)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--kernels", action="store_true")
    ap.add_argument("--syncs", action="store_true", help="list the implicit host synchronisations of one video by call site (torch.cuda.set_sync_debug_mode)")
    ap.add_argument("--cprofile", type=int, default=-1, help="cProfile the predictor call of this clip (host side)")
    ap.add_argument("--cprofile-post", default="", help="cProfile every call of this method of the loop (e.g. detect_newly_entities_per_clip_instance)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    case = cases.CFG2
    T, Q, NF = case["T"], case["Q"], args.frames
    swin = cases.build_swin(dev)
    head = cases.build_head(case, dev, return_aux=False)
    model = types.SimpleNamespace(backbone=swin, sem_seg_head=head)
    vid = synth.synthetic_frames(NF, case["H"], case["W"], "cfg3/frames").to(dev)
    loop = InferenceVideoEntity(
        hidden_dim=256, num_queries=Q, overlap_threshold_entity=0.5, stability_score_thresh=0.5, size_divisibility=32,
        pixel_mean=synth.PIXEL_MEAN, pixel_std=synth.PIXEL_STD, num_frames=T, test_topk_per_image=100, apply_cls_thres=0.25,
        box_nms_thresh=0.85, num_frames_window_test=args.window, clip_stride=1, num_prev_frames_memory=5,
        video_unified_inference_entities="", temporal_consistency_threshold=0.25, detect_newly_object_threshold=0.1,
        detect_newly_interval_frames=1, custom_videos_enable=False).to(dev)
    acc = collections.defaultdict(float)
    cnt = collections.Counter()

    def timed(obj, name, label=None):
        orig = getattr(obj, name)

        def wrapped(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = orig(*a, **k)
            torch.cuda.synchronize()
            acc[label or name] += time.perf_counter() - t0
            cnt[label or name] += 1
            return r
        setattr(obj, name, wrapped)
    for n in ("write_prompt_predictions_into_annotations_per_clip", "detect_newly_entities_per_clip_instance",
              "write_newly_entities_into_annotations_per_clip", "save_results_vis", "pad_zero_annotations_for_next_clip"):
        timed(loop, n)
    timed(model, "backbone")
    timed(head.predictor, "forward", "predictor (decoder)")
    timed(head.pixel_decoder, "forward_features", "pixel_decoder")

    def run():
        torch.manual_seed(0)
        images = normalized_image_list(list(vid), loop.pixel_mean, loop.pixel_std, 32)
        tg = [{"task": "detection", "dataset_name": "ytvis_2021_dev", "prompt_type": "visual", "num_frames": T, "video_len": NF, "sub_task": "vis"}]
        with torch.no_grad():
            loop.inference_video(model, [{"video_len": NF, "height": case["H"], "width": case["W"]}], images, tg, merge_results=False)
        return tg
    run()
    acc.clear()
    cnt.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tg = run()
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print(f"video of {NF} frames, window {args.window}: {total * 1e3:.1f} ms ({NF - T + 1} clips), entities at the end {tg[0]['ids'].shape[0] if 'ids' in tg[0] else 0}")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print(f"  {v * 1e3:9.1f} ms  {cnt[k]:3d} calls  {k}")
    print(f"  {(total - sum(acc.values())) * 1e3:9.1f} ms  everything else (slicing, class-score post-processing, loop glue)")
    if args.syncs:
        import traceback
        import warnings
        sites = collections.Counter()

        def hook(message, category, filename, lineno, file=None, line=None):
            if "synchroniz" in str(message):
                st = [f for f in traceback.extract_stack() if "univs_amd" in f.filename]
                sites[" <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:])] += 1
        old_show = warnings.showwarning
        warnings.showwarning = hook
        warnings.simplefilter("always")
        torch.cuda.set_sync_debug_mode("warn")
        run()
        torch.cuda.set_sync_debug_mode("default")
        warnings.showwarning = old_show
        print(f"implicit synchronisations of one video ({NF - T + 1} clips): {sum(sites.values())}")
        for k, v in sites.most_common(60):
            print(f"  {v:4d}  {k}")
    if args.cprofile >= 0:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        fwd = head.predictor.forward
        state = {"n": 0}

        def prof_fwd(*a, **k):
            n = state["n"]
            state["n"] += 1
            if n == args.cprofile:
                torch.cuda.synchronize()
                pr.enable()
                r = fwd(*a, **k)
                pr.disable()
                return r
            return fwd(*a, **k)
        head.predictor.forward = prof_fwd
        run()
        head.predictor.forward = fwd
        pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    if args.cprofile_post:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        orig = getattr(loop, args.cprofile_post)

        def prof_post(*a, **k):
            torch.cuda.synchronize()
            pr.enable()
            r = orig(*a, **k)
            torch.cuda.synchronize()
            pr.disable()
            return r
        setattr(loop, args.cprofile_post, prof_post)
        run()
        setattr(loop, args.cprofile_post, orig)
        pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
    if args.kernels:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            run()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=70))


if __name__ == "__main__":
    main()
