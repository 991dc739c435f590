"""Closed-form test inputs shared by oracle/gen_golden.py (which feeds them to the real reference) and
by the tests (which feed the very same tensors to the oracle and to the HIP path)."""
import math

import numpy as np
import torch

from univs_amd import synth
from univs_amd.workloads import (CFG2, CFG5, CFG5_GOLDEN_T, cfg5_frames, HEAD_CASE, HEAD_CASE_T10, PIXDEC, R50_SHAPES, SWIN_B, SWIN_L, SWIN_T, SWINL_SHAPES,  # noqa: F401
                                 SWINT_SHAPES, backbone_features, cfg2_frames, clip_table, decoder_kwargs, preprocess, sampler_kwargs,
                                 targets_first_clip, targets_with_entities)

# ---------------------------------------------------------------------------------------------------
# MSDeformAttn forward
# ---------------------------------------------------------------------------------------------------
MSDA_CASES = [
    # config-1 encoder geometry (256x448 input -> res5/res4/res3)
    dict(name="cfg1", shapes=[(8, 14), (16, 28), (32, 56)], N=2, M=8, D=32, P=4, encoder=True),
    # non-2x pyramid with odd sizes (ragged tiles, clipped windows)
    dict(name="ragged", shapes=[(5, 7), (9, 13), (17, 25)], N=1, M=8, D=32, P=4, encoder=True),
    # four levels, finest level first (tile grid must follow the largest level wherever it sits)
    dict(name="L4", shapes=[(32, 48), (16, 24), (8, 12), (4, 6)], N=1, M=8, D=32, P=4, encoder=True),
    # single level, single head
    dict(name="L1", shapes=[(20, 33)], N=2, M=1, D=32, P=4, encoder=True),
    # decoder-style: Lq != S, runtime L/P path of the vec4 kernel
    dict(name="generic", shapes=[(6, 4), (3, 2)], N=2, M=4, D=16, P=3, encoder=False, Lq=10),
    # head dim not a multiple of 4 -> scalar kernel
    dict(name="oddD", shapes=[(7, 5), (4, 3)], N=1, M=3, D=5, P=2, encoder=False, Lq=9),
]


# backward (operator boundary B2): small enough for double-precision autograd on the host
MSDA_BWD_CASES = [c for c in MSDA_CASES if c["name"] in ("ragged", "generic", "oddD")]
MSDA_BWD_CASE = MSDA_BWD_CASES[0]


def level_start_index(shapes):
    st, acc = [], 0
    for h, w in shapes:
        st.append(acc)
        acc += h * w
    return st, acc


def msda_inputs(case, dtype=torch.float32):
    """-> value [N,S,M,D], shapes list, level_start list, loc [N,Lq,M,L,P,2], attn [N,Lq,M,L,P]."""
    shapes = case["shapes"]
    N, M, D, P = case["N"], case["M"], case["D"], case["P"]
    L = len(shapes)
    lsi, S = level_start_index(shapes)
    nm = "msda/" + case["name"]
    value = synth.normal(nm + "/value", (N, S, M, D))
    if case["encoder"]:
        Lq = S
        # reference points = pixel centres of every level (msdeformattn.py:143-158, valid_ratio == 1)
        refs = []
        for (h, w) in shapes:
            ys = (torch.arange(h, dtype=torch.float32) + 0.5) / h
            xs = (torch.arange(w, dtype=torch.float32) + 0.5) / w
            yy, xx = torch.meshgrid(ys, xs, indexing="ij")
            refs.append(torch.stack([xx.reshape(-1), yy.reshape(-1)], -1))
        ref = torch.cat(refs, 0)  # [S, 2]
        off = synth.normal(nm + "/off", (N, Lq, M, L, P, 2), std=2.0)  # pixels of the target level
        # every 7th query gets far offsets (out of the tile halo, partly out of the image)
        if case.get("far", True):
            far = (torch.arange(Lq) % 7 == 3).view(1, Lq, 1, 1, 1, 1)
            off = torch.where(far, off * 6.0, off)
        norm = torch.tensor([[w, h] for (h, w) in shapes], dtype=torch.float32).view(1, 1, 1, L, 1, 2)
        loc = ref.view(1, Lq, 1, 1, 1, 2) + off / norm
        if not case.get("far", True):
            loc = loc.clamp(-0.05, 1.05)  # SURVEY.md section 8d "realistic locality" micro-benchmark input
    else:
        Lq = case["Lq"]
        loc = synth.uniform(nm + "/loc", (N, Lq, M, L, P, 2), -0.15, 1.15)
    logits = synth.normal(nm + "/attn", (N, Lq, M, L * P))
    attn = torch.softmax(logits, -1).view(N, Lq, M, L, P)
    return value.to(dtype), shapes, lsi, loc.to(dtype).contiguous(), attn.to(dtype).contiguous()


def msda_query_subset(case, Lq):
    return np.arange(0, Lq, 3)


# ---------------------------------------------------------------------------------------------------
# Swin window attention (module level: x -> qkv -> core -> proj)
# ---------------------------------------------------------------------------------------------------
WINATTN_CASES = [
    dict(name="winattn/w7_noshift", dim=96, heads=3, win=7, nW=6, batch=2, shift=False),
    dict(name="winattn/w7_shift", dim=96, heads=3, win=7, nW=6, batch=2, shift=True),
    dict(name="winattn/w12_shift", dim=128, heads=4, win=12, nW=4, batch=1, shift=True),
]


def winattn_inputs(case):
    """-> x [batch*nW, win*win, dim], mask [nW, Ntok, Ntok] or None (0 / -100, swin.py:437-440)."""
    ntok = case["win"] ** 2
    x = synth.normal(case["name"] + "/x", (case["batch"] * case["nW"], ntok, case["dim"]))
    mask = None
    if case["shift"]:
        # a plausible shifted-window mask: tokens carry a region id, different ids cannot attend
        ids = torch.floor(synth.uniform(case["name"] + "/ids", (case["nW"], ntok), 0.0, 3.0))
        ids[0] = 0.0  # first window: single region (all-zero mask row block)
        diff = ids[:, None, :] - ids[:, :, None]
        mask = torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))
    return x, mask


# ---------------------------------------------------------------------------------------------------
# Mask decode
# ---------------------------------------------------------------------------------------------------
MASKDEC_CASES = [
    dict(name="maskdec/cfg1", T=2, Q=20, C=256, H=64, W=112),
    dict(name="maskdec/ragged", T=3, Q=37, C=256, H=23, W=41),     # HW not a multiple of 32, Q not of 32
    dict(name="maskdec/bigQ", T=1, Q=150, C=256, H=16, W=28),      # more than one 128-row tile
    dict(name="maskdec/smallC", T=2, Q=5, C=64, H=8, W=9),
]


# shapes that exercise the split-bf16 kernel's corners: a ragged last 64-column tile, one / several row blocks with a
# partial last block, 100 rows (the headline query count), two row passes, a single k-step pair
MASKDEC_SPLIT_CASES = [
    dict(name="maskdec/split_q100", T=2, Q=100, C=256, H=31, W=36),    # 1116 columns = 17.4 tiles; 7 row blocks
    dict(name="maskdec/split_q106", T=1, Q=106, C=256, H=8, W=20),     # the largest single pass
    dict(name="maskdec/split_q107", T=1, Q=107, C=256, H=8, W=20),     # two passes of 54 rows
    dict(name="maskdec/split_q17", T=3, Q=17, C=128, H=10, W=14),      # 2 row blocks, the second with one row
    dict(name="maskdec/split_tiny", T=1, Q=1, C=64, H=2, W=2),         # one 4-column group
    dict(name="maskdec/split_q204", T=1, Q=204, C=256, H=46, W=80),    # config 4: 200 + 4 queries, 1/16 level
]


def maskdec_inputs(case):
    e = synth.normal(case["name"] + "/embed", (case["T"], case["Q"], case["C"]), std=0.5)
    f = synth.normal(case["name"] + "/feat", (case["T"], case["C"], case["H"], case["W"]), std=0.5)
    return e, f


# ---------------------------------------------------------------------------------------------------
# Module-level cases (Swin, pixel decoder, UniVS decoder / head)
# ---------------------------------------------------------------------------------------------------
SWIN_CASE = dict(name="swin_t", N=2, H=96, W=128)



def swin_input(case=SWIN_CASE):
    frames = synth.synthetic_frames(case["N"], case["H"], case["W"], "swin/frames")
    mean = torch.tensor(synth.PIXEL_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(synth.PIXEL_STD).view(1, 3, 1, 1)
    return (frames - mean) / std




# ---------------------------------------------------------------------------------------------------
# BASELINE config 2: Swin-T, T=5 @ 720p (padded 736x1280), 100 queries, first clip
# ---------------------------------------------------------------------------------------------------
# Swin-B (configs/univs_inf/vids/refvos/univs_swinb_refvos_davis_c1+univs.yaml:5-9): window 12 -> 144-token
# windows, the second instantiation of the window-attention kernel
SWINB_CASE = dict(name="swin_b", N=1, H=96, W=160)


# ---------------------------------------------------------------------------------------------------
# Clip loop (univs_amd/inference/video_entity.py  <->  univs/inference/inference_video_entity.py)
# ---------------------------------------------------------------------------------------------------
# (a) the real backbone + head on a 7-frame synthetic video: 3 clips of 3 frames, stride 2, window 5
LOOP_CASE = dict(name="loop", T=3, H=64, W=96, Q=20, shapes=SWINT_SHAPES, n_frames=7, image_size=(60, 90))


def loop_frames(case=LOOP_CASE):
    h, w = case["image_size"]
    return synth.synthetic_frames(case["n_frames"], h, w, case.get("frames_name", "loop/frames"))


def loop_kwargs(case=LOOP_CASE, **over):
    """Constructor arguments shared by the reference's InferenceVideoEntity and ours (the reference takes
    a few more, see oracle/gen_golden.py); thresholds of configs/univs_inf/vids/vis/Base.yaml."""
    kw = dict(hidden_dim=256, num_queries=case["Q"], overlap_threshold_entity=0.5, stability_score_thresh=0.5,
              size_divisibility=32, pixel_mean=synth.PIXEL_MEAN, pixel_std=synth.PIXEL_STD, num_frames=case["T"],
              test_topk_per_image=100, apply_cls_thres=0.25, box_nms_thresh=0.85, num_frames_window_test=5,
              clip_stride=2, num_prev_frames_memory=5, video_unified_inference_entities="",
              temporal_consistency_threshold=0.25, detect_newly_object_threshold=0.1,
              detect_newly_interval_frames=1, custom_videos_enable=False)
    kw.update(over)
    return kw


def loop_targets(case=LOOP_CASE):
    return [{"task": "detection", "dataset_name": "ytvis_2021_dev", "prompt_type": "visual", "num_frames": case["T"],
             "video_len": case["n_frames"], "sub_task": "vis"}]


def loop_batched_inputs(case=LOOP_CASE):
    return [{"video_len": case["n_frames"], "height": case["image_size"][0], "width": case["image_size"][1]}]


# (a') BASELINE config 3: a 40-frame 720p video through the sliding 5-frame clip loop (Swin-T, 100 queries); clip stride 1
# (the reference's default) -> 36 clips, window 5 (other strides make the reference's own memory-pool update raise at T=5).  The per-clip state is far too large to store at this size, so both
# sides reduce it the same way before comparing (loop_reduce).
CFG3_LOOP = dict(name="cfg3", T=5, H=736, W=1280, Q=100, shapes=SWINT_SHAPES, n_frames=40, image_size=(720, 1280),
                 reduce=True, frames_name="cfg3/frames")


def loop_reduce(tag, key, v):
    """Compact form of one state tensor of the clip loop -> {name: tensor}.  Masks become per-(entity, frame) areas,
    mask logits a 32x32-strided sample of the newest frames plus the count of near-zero logits per (entity, frame) (the
    slack the area comparison is allowed), the prompt memory a strided sample."""
    v = v.detach()
    if key == "masks":
        return {f"{tag}_masks_area": v.flatten(-2).double().sum(-1).long()}
    if key == "mask_logits":
        return {f"{tag}_mask_logits_s": v[:, -5:, ::32, ::32].float().clone(),
                f"{tag}_mask_logits_near0": (v.abs() < 1e-3).flatten(-2).sum(-1).long()}
    if key == "logits":     # [N_ent, frames, 3938 classes]: every 16th class + the winner
        return {f"{tag}_logits_s": v[..., ::16].float().clone(), f"{tag}_logits_max": v.max(-1).values.float(),
                f"{tag}_logits_argmax": v.argmax(-1).long()}
    if key in ("prompt_pe", "prompt_feats"):
        return {f"{tag}_{key}_s": v[:, ::16, :, ::16].float().clone()}
    if key == "prompt_attn_masks":
        return {f"{tag}_prompt_attn_masks_frac": v.float().mean(-1)}
    return {f"{tag}_{key}": v.float().clone() if v.dtype == torch.bool else v.clone()}


# (b) a scripted scene instead of the network: rectangles that move, appear, leave and duplicate each
# other, so that every branch of the per-clip bookkeeping runs with several entities.  Used as
# `model.sem_seg_head` by BOTH the reference loop (golden generation) and ours.
SCRIPT_CASE = dict(name="script", T=3, H=64, W=96, Q=8, K=16, n_frames=7, image_size=(60, 90))
# class, class logit, x0, y0, w, h, vx, vy, first frame, last frame
SCRIPT_OBJECTS = [
    (3, 3.0, 4, 6, 30, 24, 3, 1, 0, 6),
    (5, 2.5, 50, 30, 28, 22, 0, 0, 0, 6),
    (5, 2.0, 50, 30, 28, 22, 0, 0, 0, 6),     # duplicate of object 1 with a lower score (box NMS)
    (9, 3.0, 4, 40, 24, 18, 1, 0, 3, 6),      # appears at frame 3 -> new entity on the second clip
    (1, 2.8, 70, 2, 18, 20, 0, 1, 0, 2),      # leaves after frame 2
    (7, -1.2, 40, 2, 20, 14, 1, 0, 0, 6),     # never confident enough
]


# edge scenes: nothing is ever confident / the first entity shows up only in the third clip and everything else is gone
SCRIPT_SCENES = {
    "empty": [(7, -1.2, 40, 2, 20, 14, 1, 0, 0, 6), (2, -2.0, 10, 30, 20, 14, 0, 0, 0, 6)],
    "late": [(3, 3.0, 4, 6, 30, 24, 3, 1, 4, 6), (1, 2.8, 60, 30, 18, 20, 0, 0, 5, 6)],
    "leavers": [(3, 3.0, 4, 6, 30, 24, 3, 1, 0, 1), (5, 2.5, 50, 30, 28, 22, 0, 0, 0, 2)],
}


class ScriptedBackbone:
    def __call__(self, x):
        return {"res2": x.new_zeros((x.shape[0], 1, 1, 1))}


class ScriptedHead:
    """Deterministic stand-in for `sem_seg_head(features, targets=targets)`: learnable query j reports
    object j; one prompt query per entity of the pool reports the object that entity has been following
    (largest overlap with its stored masks)."""

    def __init__(self, case=SCRIPT_CASE, objects=SCRIPT_OBJECTS):
        self.case, self.objects = case, objects
        C = 256
        self.emb = [torch.nn.functional.normalize(synth.normal(f"script/emb/{j}", (C,)), dim=0) * 8.0
                    for j in range(len(objects))]
        self.space = 0.3 * synth.normal("script/space", (C,))

    def rect(self, j, f):
        _, _, x0, y0, w, h, vx, vy, f0, f1 = self.objects[j]
        if f < f0 or f > f1:
            return None
        return x0 + vx * f, y0 + vy * f, x0 + vx * f + w, y0 + vy * f + h

    def mask_logits(self, j, f, stride=4):
        H, W = self.case["H"] // stride, self.case["W"] // stride
        r = None if j is None else self.rect(j, f)
        if r is None:
            return torch.full((H, W), -6.0)
        ys = (torch.arange(H).float() * stride + stride / 2).view(-1, 1)
        xs = (torch.arange(W).float() * stride + stride / 2).view(1, -1)
        d = torch.maximum(torch.maximum(r[0] - xs, xs - r[2]), torch.maximum(r[1] - ys, ys - r[3]))
        return (-0.75 * d).clamp(-6.0, 6.0)

    def followed_object(self, masks, frame0=0):
        """masks [t_hist, H, W] of one entity (masks[0] = absolute frame `frame0`) -> index of the object it overlaps
        most (None if none)."""
        best, best_j = 0.0, None
        for j in range(len(self.objects)):
            ov = 0.0
            for f0 in range(masks.shape[0]):
                f = f0 + frame0
                r = self.rect(j, f)
                if r is not None:
                    ov += float(masks[f0, max(r[1], 0):max(r[3], 0), max(r[0], 0):max(r[2], 0)].float().sum())
            if ov > best:
                best, best_j = ov, j
        return best_j

    def row(self, j, frames, kind):
        K = self.case["K"]
        visible = j is not None and any(self.rect(j, f) is not None for f in frames)
        tag = f"script/{kind}/{j}/{frames[0]}"
        logits = torch.full((K,), -4.0) + 0.05 * synth.uniform(tag + "/cls", (K,))
        if visible:
            logits[self.objects[j][0]] = self.objects[j][1]
        masks = torch.stack([self.mask_logits(j if visible else None, f) for f in frames])
        noise = synth.normal(tag + "/emb", (len(frames), 256))
        if visible:
            embds = self.emb[j][None] + 0.1 * noise + (self.space[None] if kind == "p" else 0.0)
        else:
            embds = 0.5 * noise
        return logits, masks, embds

    def __call__(self, features, targets=None):
        tv = targets[0]
        frames = [int(f) for f in tv["frame_indices"]]
        dev = features["res2"].device
        rows = [self.row(j if j < len(self.objects) else None, frames, "l") for j in range(self.case["Q"])]
        if "masks" in tv:
            frame0 = frames[-1] + 1 - tv["masks"].shape[1]       # absolute frame of masks[:, 0]
            for e in range(tv["masks"].shape[0]):
                if tv.get("task") == "grounding":                # expression e refers to object exp_obj_ids[e]
                    j = int(tv["exp_obj_ids"][e])
                else:
                    j = self.followed_object(tv["masks"][e].cpu(), frame0)
                rows.append(self.row(j, frames, "p"))
        return {"pred_logits": torch.stack([r[0] for r in rows])[None].to(dev),
                "pred_masks": torch.stack([r[1] for r in rows])[None].to(dev),
                "pred_embds": torch.stack([r[2] for r in rows])[None].to(dev),
                "aux_outputs": []}


# ---------------------------------------------------------------------------------------------------
# BASELINE config 4: Swin-B (window 12), T=5 @ 720p, 200 learnable queries + 4 referring expressions
# (grounding, 'sep-blocked' self-attention mask, text prompts fused into the image features)
# ---------------------------------------------------------------------------------------------------
from univs_amd.workloads import CFG4, CFG4_DECODER, SWINB_SHAPES, cfg4_targets, targets_grounding  # noqa: E402,F401


# panoptic sub-task on the scripted scene: categories (1-based) of objects 0, 3, 4 are "things", the others "stuff"
SCRIPT_THING_IDS = (4, 10, 2)


# (c) VOS / RefVOS drivers (univs_amd/inference/video_vos.py <-> univs/inference/inference_video_vos.py) on the scripted
# scene: objects 0 and 1 are annotated in frame 0, object 3 enters (and is annotated) in frame 3
VOS_OBJECTS = {11: (0, 0), 22: (1, 0), 33: (3, 3)}      # original id -> (scripted object, annotated frame)


def vos_kwargs(case=SCRIPT_CASE, **over):
    kw = dict(hidden_dim=256, num_queries=case["Q"], size_divisibility=32, pixel_mean=synth.PIXEL_MEAN,
              pixel_std=synth.PIXEL_STD, num_frames=case["T"], prompt_as_queries=True, num_frames_window_test=5,
              clip_stride=2, video_unified_inference_queries="prompt", num_prev_frames_memory=5)
    kw.update(over)
    return kw


def vos_targets_sot(make_annotations, case=SCRIPT_CASE, objects=None, class_offset=0, dataset="ytbvos18_val"):
    """`make_annotations(image_size, ori_ids, gt_masks, gt_boxes, gt_classes)` builds one frame's annotation object
    (our FrameAnnotations or a stand-in for detectron2 Instances in the golden generator)."""
    objects = SCRIPT_OBJECTS if objects is None else objects
    head = ScriptedHead(case, objects)
    h, w = case["image_size"]
    per_frame = []
    for f in range(case["n_frames"]):
        ids, masks, boxes, classes = [], [], [], []
        for oid, (j, f_ann) in VOS_OBJECTS.items():
            if f_ann == f:
                x0, y0, x1, y1 = head.rect(j, f)
                m = torch.zeros(h, w)
                m[y0:y1, x0:x1] = 1.0
                ids.append(oid); masks.append(m); boxes.append(torch.tensor([x0, y0, x1, y1], dtype=torch.float32))
                classes.append(objects[j][0] - class_offset)
        per_frame.append(make_annotations((h, w), ids, torch.stack(masks) if masks else torch.zeros(0, h, w),
                                          torch.stack(boxes) if boxes else torch.zeros(0, 4),
                                          torch.tensor(classes, dtype=torch.long)))
    names = [f"videos/clip0/{f:05d}.jpg" for f in range(case["n_frames"])]
    return [{"task": "sot", "dataset_name": dataset, "prompt_type": "visual", "num_frames": case["T"],
             "video_len": case["n_frames"], "inter_image_size": (case["H"], case["W"]), "image_size": (h, w),
             "instances": per_frame, "file_names": names, "mask_palette": [0] * 768}]


def vos_targets_grounding(case=SCRIPT_CASE):
    h, w = case["image_size"]
    names = [f"videos/clip0/{f:05d}.jpg" for f in range(case["n_frames"])]
    return [{"task": "grounding", "dataset_name": "rvos-refytb-val", "prompt_type": "text", "num_frames": case["T"],
             "video_len": case["n_frames"], "inter_image_size": (case["H"], case["W"]), "image_size": (h, w),
             "exp_obj_ids": [0, 3, 4], "file_names": names}]


# (d) CLIP text tower (univs_amd/modeling/language <-> univs/modeling/language): a tiny geometry and the RN50x4 one
TEXT_SMALL = dict(embed_dim=48, context_length=77, vocab_size=49408, transformer_width=64, transformer_heads=4,
                  transformer_layers=3)
TEXT_FULL = dict(embed_dim=640, context_length=77, vocab_size=49408, transformer_width=640, transformer_heads=10,
                 transformer_layers=12)


def build_text_encoder(cfg, device="cpu"):
    from univs_amd.modeling.language import CLIPLangEncoder
    return synth.load_synthetic(CLIPLangEncoder(**cfg), "lang_encoder.").eval().to(device)


# (e) result formats: per-clip records of three entities over a 9-frame video (clips of 3 frames, 5 classes); entity 7 is
# blank in the middle clip, entity 9 only exists in the last clip, entity 3 carries mask-quality scores
def vis_result_records(height=12, width=10):
    def masks(tag, t):
        return synth.uniform(f"results/{tag}", (t, height, width)) > 0.2
    def score(tag, scale=1.0):
        return (synth.uniform(f"results/score/{tag}", (5,)) * 0.5 + 0.5) * scale
    clips = [
        [dict(obj_id=3, score=score("3a"), masks=masks("3a", 3), frame_id_start=0),
         dict(obj_id=7, score=score("7a", 0.04), masks=masks("7a", 3), frame_id_start=0)],
        [dict(obj_id=3, score=score("3b"), masks=masks("3b", 3), frame_id_start=3),
         dict(obj_id=7, score=torch.zeros(5), masks=torch.zeros(3, height, width, dtype=torch.bool), frame_id_start=3)],
        [dict(obj_id=3, score=score("3c"), masks=masks("3c", 3), frame_id_start=6, mask_quality_score=torch.tensor(0.8)),
         dict(obj_id=7, score=score("7c", 0.04), masks=masks("7c", 3), frame_id_start=6, mask_quality_score=torch.tensor(0.5)),
         dict(obj_id=9, score=score("9c", 0.3), masks=masks("9c", 2), frame_id_start=7)],
    ]
    info = [{"video_id": "17", "video_len": 9, "height": height, "width": width}]
    return info, clips


# (f) caller-side target preparation (univs_amd/prepare_targets.py <-> univs/prepare_targets.py: process_inference)
def prepare_targets_inputs():
    """scenario -> (constructor overrides, batched_inputs as the dataset mapper hands them over)"""
    names = [f"videos/clipA/{f:05d}.jpg" for f in range(4)]
    base = dict(video_len=4, file_names=names)
    return {
        "detection_known": ({}, [dict(base, task="detection", dataset_name="ytvis_2021", video_id=17, is_raw_video=False)]),
        "detection_raw": ({}, [dict(base, task="detection", dataset_name="my_videos", is_raw_video=True)]),
        "detection_semantic": (dict(semantic_on=True), [dict(base, task="detection", dataset_name="vipseg", is_raw_video=False)]),
        "detection_vspw": ({}, [dict(base, task="detection", dataset_name="vspw_vss_video_val", is_raw_video=False)]),
        "grounding": ({}, [dict(base, task="grounding", dataset_name="rvos-refytb-val", expressions=["a dog running", "the person on the left"],
                                exp_obj_ids=[3, 7])]),
        "grounding_empty": ({}, [dict(base, task="grounding", dataset_name="rvos-refytb-val", expressions=[], exp_obj_ids=[])]),
        "sot": ({}, [dict(base, task="sot", dataset_name="ytbvos18_val", instances=["f0", "f1", "f2", "f3"], mask_palette=[1, 2, 3],
                          video_id="ab12")]),
        "custom_text": (dict(custom_videos_text=[["two zebras", "a red car"]]),
                        [dict(base, task="detection", dataset_name="my_videos", is_raw_video=True)]),
    }


# (g) VIPOSeg (panoptic VOS): class scores live in the VIPSeg slice of the 3938-wide class table (start 2924); classes 5
# and 9 are 'stuff' (dataset ids 6 and 10), so objects 1 / 2 (both class 5) and 3 borrow pixels from the semantic map
VIPOSEG_CASE = dict(SCRIPT_CASE, name="viposeg", K=3938)
VIPOSEG_CLASS_START = 2924
VIPOSEG_OBJECTS = [(VIPOSEG_CLASS_START + o[0],) + tuple(o[1:]) for o in SCRIPT_OBJECTS]
VIPOSEG_STUFF_IDS = (6, 10)


# ---------------------------------------------------------------------------------------------------
# (d) on-disk VPS / VSS result formats (univs_amd/inference/results.py <-> univs/evaluation/vps_evaluation.py, vss_evaluation.py)
# ---------------------------------------------------------------------------------------------------
VPS_CATEGORIES = {1: {"id": 1, "isthing": 0, "color": [120, 120, 120]}, 2: {"id": 2, "isthing": 1, "color": [180, 120, 120]},
                  3: {"id": 3, "isthing": 1, "color": [6, 230, 230]}, 4: {"id": 4, "isthing": 0, "color": [80, 50, 50]},
                  7: {"id": 7, "isthing": 1, "color": [4, 200, 3]}}


def result_file_inputs(T=3):
    return {"video_id": "vid_0007", "file_names": [f"datasets/vipseg/imgs/vid_0007/{10 + 2 * t:08d}.jpg" for t in range(T + 1)],
            "frame_indices": list(range(T))}


def vps_result_outputs(T=3, H=24, W=40):
    """A scripted panoptic result: two things of one class (the second needs a random colour), a thing that leaves, a stuff region, an
    id that never appears, unlabelled pixels (0)."""
    pan = torch.zeros(T, H, W, dtype=torch.int64)
    pan[:, :8] = 90                                  # stuff, class 1
    pan[:, 10:20, 3:12] = 31                         # thing, class 2
    pan[:, 12:22, 20:33] = 32                        # thing, class 2 again
    pan[0, 2:6, 30:38] = 33                          # thing, class 7, first frame only
    pan[1:, 20:24, 0:6] = 91                         # stuff, class 4, from the second frame on
    infos = [{"id": 31, "isthing": True, "category_id": 2}, {"id": 32, "isthing": True, "category_id": 2},
             {"id": 33, "isthing": True, "category_id": 7}, {"id": 34, "isthing": True, "category_id": 3},
             {"id": 90, "isthing": False, "category_id": 1}, {"id": 91, "isthing": False, "category_id": 4}]
    return {"image_size": (H, W), "pred_masks": pan, "segments_infos": infos, "task": "vps"}


VSS_CONTIGUOUS_TO_DATASET = {0: 1, 1: 2, 2: 5, 3: 9}


def vss_result_outputs(T=3, H=24, W=40):
    sem = torch.zeros(T, H, W, dtype=torch.int64)
    sem[:, 6:, :] = 2
    sem[1:, 10:18, 8:30] = 3
    sem[0, :3, :5] = 255                             # the ignore value
    sem[2, 20:, 30:] = 1
    return {"image_size": (H, W), "pred_masks": sem, "task": "vss"}
