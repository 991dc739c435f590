"""Clip loop for entity segmentation in videos (VIS-style sub-tasks) around the hot path.

Counterpart of the reference's `InferenceVideoEntity` (univs/inference/inference_video_entity.py):
the same method names, argument meaning and -- above all -- the same mutations of the per-video
`targets[0]` dictionary, which is the state the prompt-as-query decoder reads on the next clip
(`masks`, `boxes`, `ids`, `first_appear_frame_idxs`, the `prompt_*` memory pool, ...).

    eval / inference_video          :236-431   windowed backbone, stride rule, clip schedule
    write_prompt_predictions_into_annotations_per_clip   :432-513
    detect_newly_entities_per_clip_instance              :515-651
    write_newly_entities_into_annotations_per_clip       :767-877
    pad_zero_annotations_for_next_clip                   :879-912
    save_results_vis                                     :914-960 (masks returned as bool tensors; the
                                                          COCO-RLE json is a result *format*)

Everything stays on the device of the inputs; the only host round trip per clip is the Hungarian
solve on the [entities x queries] similarity matrix (as in the reference, scipy).  The per-query
Python loop of the reference's new-entity test (:640-645) is one batched mask-IoU here.

Sub-tasks: 'vis' / 'entity_vis_*' (instance-style), 'vps' (panoptic: things + stuff, segment ids remembered in
`targets[0]`) and 'vss' (semantic: per-clip class x mask maps, no entity bookkeeping).
"""
import math
from typing import Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ..registry import configurable
from ..utils.memory import retry_if_oom
from ..layers import to_device_async
from ..utils.comm import batched_mask_iou, calculate_mask_quality_scores, convert_mask_to_box, count_true, video_box_iou
from .comm import check_consistency_with_prev_frames, match_from_learnable_embds  # noqa: F401  (API parity)
from scipy.optimize import linear_sum_assignment

# (num_classes, start index) of each dataset inside the concatenated CLIP class-embedding table: the
# layout of the reference's `clip_class_embed_path` file (data, restated from
# datasets/concept_emb/combined_datasets_category_info.py:7-24; 3938 rows in total)
COMBINED_DATASETS_CATEGORY_INFO = {
    "imagenet": (1000, 0), "lvis": (1203, 1000), "burst": (1203, 1000), "ytvis21": (40, 2203), "ovis": (25, 2243),
    "bdd_track": (8, 2268), "objects365": (365, 2276), "coco_panoptic": (133, 2641), "coco": (80, 2641),
    "ade20k": (150, 2774), "vipseg": (124, 2924), "vspw": (124, 2924), "viposeg": (124, 2924), "ytvis19": (40, 3048),
    "entityseg_instance": (206, 3088), "entityseg_panoptic": (644, 3294),
}
ENTITY_SUBTASK_DATASET = {   # inference_video_entity.py:320-333
    "entity_vss_entityseg": "entityseg_panoptic",
    "entity_vps_entityseg": "entityseg_panoptic",
    "entity_vss_vipseg": "vipseg",
    "entity_vps_vipseg": "vipseg",
    "entity_vis_entityseg": "entityseg_instance",
    "entity_vis_coco": "coco",
}


class ImageList:
    """Minimal stand-in for detectron2.structures.ImageList: `.tensor` [N, C, Hp, Wp] zero-padded to a
    multiple of `size_divisibility`, `.image_sizes` the unpadded (H, W) per image."""

    def __init__(self, tensor, image_sizes):
        self.tensor, self.image_sizes = tensor, image_sizes

    @staticmethod
    def from_tensors(tensors, size_divisibility=0):
        sizes = [tuple(t.shape[-2:]) for t in tensors]
        H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)
        if size_divisibility > 1:
            d = size_divisibility
            H, W = (H + d - 1) // d * d, (W + d - 1) // d * d
        out = tensors[0].new_zeros((len(tensors),) + tuple(tensors[0].shape[:-2]) + (H, W))
        for i, t in enumerate(tensors):
            out[i, ..., : t.shape[-2], : t.shape[-1]] = t
        return ImageList(out, sizes)


def normalized_image_list(frames, pixel_mean, pixel_std, size_divisibility):
    """`ImageList.from_tensors([(f - mean) / std for f in frames], size_divisibility)` (inference_video_entity.py:246-250).  Frames of
    one size on the GPU take ONE pass (ops.normalize_pad: bit-identical to the subtraction, the division and the padded copy)."""
    if (len(frames) > 0 and frames[0].is_cuda and frames[0].dtype == torch.float32 and frames[0].dim() == 3
            and all(f.shape == frames[0].shape for f in frames) and not torch.is_grad_enabled()):
        from .. import ops
        out = ops.normalize_pad(torch.stack(frames), pixel_mean, pixel_std, size_divisibility)
        if out is not None:
            return ImageList(out, [tuple(f.shape[-2:]) for f in frames])
    return ImageList.from_tensors([(f - pixel_mean) / pixel_std for f in frames], size_divisibility)


def _resize(masks, size):
    # (the reference wraps these resizes in retry_if_cuda_oom, inference_video_entity.py:933 / :978 / :1104: utils/memory.py)
    return retry_if_oom(F.interpolate)(masks, size, mode="bilinear", align_corners=False)


_NORM4 = {}


def _norm4(w, h, device):
    """[w, h, w, h] on `device`, made once per size: `torch.as_tensor(list, device=...)` is a pageable host-to-device copy, which waits
    for everything enqueued on the stream before it (hazard 14) -- a hidden synchronisation per call."""
    key = (int(w), int(h), str(device))
    t = _NORM4.get(key)
    if t is None:
        if len(_NORM4) > 64:
            _NORM4.clear()
        t = _NORM4[key] = torch.as_tensor([w, h, w, h], device=device)
    return t


def _plane_stats(x, valid=None):
    """Logits x [N, T, H, W] -> integer statistics per plane [N, T, 8]: (|{x > 1}|, |{x > -1}|, left, top, right, bottom of {x > 0} --
    inclusive, zeros when empty --, non-empty, 0) over rows / columns below `valid` (the whole plane by default).  On the GPU one pass of
    `ops.mask_stats` (csrc/mask_stats.hip) instead of ~25 launches; elsewhere (and for shapes it does not cover) the ATen formulation of
    `calculate_mask_quality_scores` / `convert_mask_to_box`, packed the same way."""
    if x.is_cuda and x.dtype == torch.float32 and x.numel() > 0 and (x.is_contiguous() or (
            x.dim() == 4 and x.stride(-1) == 1 and x.stride(-2) == x.shape[-1] and x.stride(1) >= x.shape[-2] * x.shape[-1])):
        from .. import ops
        st = ops.mask_stats(x, 1.0, -1.0, 0.0, valid=valid)      # (a [N, T, H, W] view of a longer history is read in place)
        if st is not None:
            return st
    cur = x if valid is None else x[..., : valid[0], : valid[1]]
    hi = (cur > 1.0).flatten(-2).sum(-1)
    lo = (cur > -1.0).flatten(-2).sum(-1)
    fg = cur > 0
    box = convert_mask_to_box(fg).to(hi.dtype)
    ne = fg.flatten(-2).any(-1).to(hi.dtype)
    return torch.cat([hi[..., None], lo[..., None], box, ne[..., None], torch.zeros_like(hi)[..., None]], -1)


def _quality_counts_and_boxes(x, valid=None, boxes=True, stats=None):
    """Logits x [N, T, H, W] (or their `_plane_stats`) -> (|{x > 1}| per entity [N] int64, |{x > -1}| per entity clamped to >= 1 [N]
    int64 -- the two counts of `calculate_mask_quality_scores` over rows / columns below `valid` --, integer boxes of {x > 0} [N, T, 4]
    int64 over the WHOLE plane as `convert_mask_to_box` returns them, or None)."""
    st = _plane_stats(x, valid) if stats is None else stats
    hi = st[..., 0].sum(-1)
    lo = st[..., 1].sum(-1).clamp(min=1)
    bx = None
    if boxes:
        full = stats is not None or valid is None or (valid[0] >= x.shape[-2] and valid[1] >= x.shape[-1])
        bx = (st if full else _plane_stats(x))[..., 2:6].long()
    return hi, lo, bx


def _refresh_recent_masks(tv, T):
    """`tv["masks"] = tv["mask_logits"].gt(0).float()` (:476, :634) when only the last T frames of the logits changed: the binarised
    history is kept equal to the logits' sign by everything that touches it (padding, newcomers, the output window), so the earlier
    frames need no second pass over `[N, history, H, W]` (0.75 GB at ten entities and ten frames of history).  The first clip stores
    the masks as bool (write_newly_entities_into_annotations_per_clip, as the reference does): then the whole tensor is rebuilt."""
    ml, mk = tv["mask_logits"], tv["masks"]
    if mk.dtype == torch.float32 and mk.shape == ml.shape and mk.device == ml.device and 0 < T <= ml.shape[1]:
        mk[:, -T:] = ml[:, -T:].gt(0.0)
    else:
        tv["masks"] = ml.gt(0.0).float()


class ClipMaskRows:
    """The mask logits of a clip that STAY on the ranks that computed them (frame-sharded clip loop; SURVEY 8e: masks kept sharded, counts
    and boxes exchanged): the book-keeping asks for per-plane statistics of all rows (tiny), for the full-clip logits of the FEW rows that
    enter the replicated per-video state (prompt queries, matched and new entities) and for the mask IoU of candidate rows against the
    known entities (evaluated per frame where the frame lives, reduced by a maximum) -- instead of every rank receiving every row.
    Every method is a collective over the loop's group: all ranks call it with the same (replicated) arguments; ranks outside the clip's
    team (more ranks than frames) receive the result from the team's first rank."""

    def __init__(self, shard, team, cs, local, n_rows, n_frames, hw, device):
        self.shard, self.team, self.cs, self.local = shard, team, cs, local          # cs / local: None outside the team
        self.n_rows, self.T, self.hw, self.device = int(n_rows), int(n_frames), (int(hw[0]), int(hw[1])), device
        self.whole = len(team) == shard.world

    def _account(self, kind, before):
        b = self.shard.bytes
        b["result:" + kind] += b[kind] - before
        b[kind] = before

    def _to_outside(self, t, shape, dtype):
        if self.whole:
            return t
        import torch.distributed as dist
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=self.device)
        t = t.contiguous()
        if t.numel():
            dist.broadcast(t, src=self.shard.global_rank(self.team[0]), group=self.shard.group)
            if self.shard.rank != self.team[0]:
                self.shard.bytes["result:broadcast"] += t.numel() * t.element_size()
        return t

    def _index(self, idx):
        """Row indices as a long tensor on the masks' device (host indices through pinned staging: hazard 14)."""
        if isinstance(idx, torch.Tensor) and idx.device == self.device:
            return idx
        return to_device_async(torch.as_tensor(idx, dtype=torch.long).cpu(), self.device)

    def stats(self):
        """`_plane_stats` of every row and frame of the clip: [n_rows, T, 8]."""
        full = None
        if self.cs is not None:
            st = _plane_stats(self.local).to(torch.int32)
            before = self.shard.bytes["all_gather"]
            full = self.cs.all_gather_frames(st, dim=1)
            self._account("all_gather", before)
        return self._to_outside(full, (self.n_rows, self.T, 8), torch.int32)

    def rows(self, idx):
        """Full-clip logits [len(idx), T, h, w] of the rows `idx`."""
        n = len(idx)
        if n == 0:
            return torch.zeros((0, self.T) + self.hw, device=self.device)
        full = None
        if self.cs is not None:
            before = self.shard.bytes["all_gather"]
            full = self.cs.all_gather_frames(self.local.index_select(0, self._index(idx)), dim=1)
            self._account("all_gather", before)
        return self._to_outside(full, (n, self.T) + self.hw, torch.float32)

    def iou_max(self, idx, known):
        """max over the clip's frames and the known entities of the mask IoU of rows `idx` ({logit > 0}) with `known` [T, N, h, w] bool
        (replicated): [len(idx)] float32."""
        n = len(idx)
        v = None
        if self.cs is not None:
            pos = torch.as_tensor(self.cs.local_positions, device=known.device)
            m = self.local.index_select(0, self._index(idx)).transpose(0, 1).gt(0.0)          # [T_loc, n, h, w]
            v = batched_mask_iou(m, known.index_select(0, pos)).amax(dim=(0, 2))
            before = self.shard.bytes["all_reduce"]
            v = self.cs.all_reduce_max(v)
            self._account("all_reduce", before)
        return self._to_outside(v, (n,), torch.float32)


def window_features_on_owner(model, x, frames, shard):
    """backbone + pixel decoder of `frames` (absolute indices of the video) on the frames THIS rank owns (frame f belongs to rank
    f % world) -> ({frame: row}, (mask_features, mask_features_bfe_conv, multi_scale_features) of those rows)."""
    mine = [f for f in frames if f % shard.world == shard.rank]
    if not mine:
        return {}, None
    feats = model.backbone(x[mine] if mine != list(range(mine[0], mine[0] + len(mine))) else x[mine[0]:mine[0] + len(mine)])
    mf, bfe, _enc, ms = model.sem_seg_head.pixel_decoder.forward_features(feats)
    return {f: k for k, f in enumerate(mine)}, (mf, bfe, list(ms))


POOL_KEYS = ("prompt_pe", "prompt_feats", "prompt_attn_masks")     # the prompt memory pool in targets[0] (modeling/prompt_encoder.py)


def _sampler_encoder(model):
    try:
        return model.sem_seg_head.predictor.visual_prompt_sampler.visual_prompt_encoder
    except AttributeError:
        return None


def sharded_clip_forward(model, targets, first, n_clip, rows, pd, shard, device=None, lazy_masks=False):
    """The head's predictor on the clip [first, first + n_clip) whose frames are spread over the ranks of `shard` (ClipShard: one
    all-gather of the query states per decoder layer) -> the full-clip output dict on every rank (mask logits and embeddings of the
    clip's frames all-gathered; class / re-id logits are replicated by construction).
    MORE RANKS THAN FRAMES (an 8-GPU node on 5-frame clips): the decoder runs on the TEAM of ranks that own a frame of this clip -- a
    sub-group, made once per distinct team -- while the others have nothing to do for this clip (their share of the node's work is the
    backbone + pixel decoder of the frames they own, `window_features_on_owner`); the clip's outputs, the prompt memory pool and the
    state of the random generators then go from the team's first rank to the ranks outside it, so that the replicated per-video
    state stays identical everywhere (every rank is in the team of some later clip).
    `lazy_masks`: returns (out, ClipMaskRows) -- the mask logits are NOT gathered: `out["pred_masks"]` holds this rank's frames (None
    outside the team) and the book-keeping asks the ClipMaskRows for what it needs (the 'vis' loop does)."""
    from ..distributed import ClipShard, cyclic_owners
    predictor = model.sem_seg_head.predictor
    if getattr(predictor, "semantic_extraction_enable", False):
        # (that mode returns pred_embds as [T_loc, C, Q'] and a per-rank `mask_features`: neither is gathered here -- ADVICE r05)
        raise NotImplementedError("frame-sharded clip loop: MODEL.UniVS.TEST.SEMANTIC_EXTRACTION.ENABLE is not supported")
    owners = cyclic_owners(first, n_clip, shard.world)
    team = sorted(set(owners))
    whole = len(team) == shard.world
    group = shard.group if whole else shard.subgroup(team)
    tv = targets[0]
    out = None
    if shard.rank in team:
        cs = ClipShard([team.index(o) for o in owners], group=group, always_collective=shard.always_collective, counter=shard.bytes)
        k = [rows[first + p] for p in cs.local_positions]
        sel = (lambda t: t[k[0]:k[0] + len(k)]) if k == list(range(k[0], k[0] + len(k))) else (lambda t: t[k])
        mf, bfe, ms = pd
        predictor.frame_shard = cs
        try:
            out = predictor([sel(lv) for lv in ms], sel(mf), sel(bfe) if bfe is not None else None, None, targets)
        finally:
            predictor.frame_shard = None
        before = shard.bytes["all_gather"]
        if not lazy_masks:
            out["pred_masks"] = cs.all_gather_frames(out["pred_masks"], dim=2)     # [1, Q', T_loc, h, w] -> [1, Q', T, h, w]
        out["pred_embds"] = cs.all_gather_frames(out["pred_embds"], dim=2)         # [1, Q', T_loc, C]
        shard.bytes["result:all_gather"] += shard.bytes["all_gather"] - before
        shard.bytes["all_gather"] = before
        if lazy_masks:
            pm = out["pred_masks"]
            lazy = ClipMaskRows(shard, team, cs, pm[0].float().contiguous(), pm.shape[1], n_clip, pm.shape[-2:], pm.device)
    if whole:
        return (out, lazy) if lazy_masks else out
    # ---- the ranks outside the team: outputs, memory pool, generator states from the team's first rank
    if device is None:
        device = pd[0].device if pd is not None else torch.device("cpu")
    enc = _sampler_encoder(model)
    src = team[0]
    state = {}
    if shard.rank == src:
        state = {f"out:{k_}": v for k_, v in out.items() if isinstance(v, torch.Tensor) and not (lazy_masks and k_ == "pred_masks")}
        if lazy_masks:
            state["masks_shape"] = torch.tensor([out["pred_masks"].shape[1], out["pred_masks"].shape[-2], out["pred_masks"].shape[-1]])
        state.update({f"pool:{k_}": tv[k_] for k_ in POOL_KEYS if k_ in tv})
        state["rng:cpu"] = torch.get_rng_state()
        if enc is not None and str(device) in enc._dev_gen:           # (the device generator of the sampler's "device" mode)
            state["rng:dev"] = enc._dev_gen[str(device)].get_state()
    before = shard.bytes["broadcast"]
    got = shard.broadcast_state(src, state, device)
    shard.bytes["result:broadcast"] += shard.bytes["broadcast"] - before
    shard.bytes["broadcast"] = before
    if shard.rank not in team:
        out = {k_[4:]: v for k_, v in got.items() if k_.startswith("out:")}
        out.setdefault("pred_reid_logits", None)
        for k_, v in got.items():
            if k_.startswith("pool:"):
                tv[k_[5:]] = v
        if any(k_.startswith("pool:") for k_ in got):
            tv["prompt_obj_ids"] = tv["ids"]
        torch.set_rng_state(got["rng:cpu"])
        if enc is not None and "rng:dev" in got:
            enc._generator(device).set_state(got["rng:dev"])
        if lazy_masks:
            out["pred_masks"] = None
            qp, mh, mw = (int(v) for v in got["masks_shape"].tolist())
            lazy = ClipMaskRows(shard, team, None, None, qp, n_clip, (mh, mw), device)
    return (out, lazy) if lazy_masks else out


def begin_video(model, device, shard):
    """The prompt sampler's per-video (re)seeding: modeling/prompt_encoder.py: VisualPromptEncoder.begin_video."""
    try:
        enc = model.sem_seg_head.predictor.visual_prompt_sampler.visual_prompt_encoder
    except AttributeError:      # (a caller's head without our predictor: nothing to seed)
        return None
    return enc.begin_video(device, shard)


def check_loop_shard(shard, num_frames):
    """None for a one-rank shard without forced collectives.  (A group larger than a clip is fine: sharded_clip_forward runs every
    clip's decoder on the ranks that own one of its frames.)"""
    if shard is not None and shard.world == 1 and not shard.always_collective:
        return None
    return shard


class InferenceVideoEntity(nn.Module):
    @configurable
    def __init__(
        self,
        *,
        hidden_dim: int,
        num_queries: int,
        overlap_threshold_entity: float,
        stability_score_thresh: float,
        size_divisibility: int,
        pixel_mean: Tuple[float],
        pixel_std: Tuple[float],
        num_frames: int,
        test_topk_per_image: int,
        apply_cls_thres: float,
        box_nms_thresh: float,
        num_frames_window_test: int,
        clip_stride: int,
        num_prev_frames_memory: int = 5,
        video_unified_inference_entities: str = "",
        temporal_consistency_threshold: float = 0.5,
        detect_newly_object_threshold: float = 0.05,
        detect_newly_interval_frames: int = 1,
        custom_videos_enable: bool = False,
        dataset_category_info=None,
        overlap_threshold: float = 0.0,
        thing_dataset_ids=(),
    ):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.num_queries = num_queries
        self.overlap_threshold_entity = overlap_threshold_entity
        self.stability_score_thresh = stability_score_thresh
        self.size_divisibility = size_divisibility
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean, dtype=torch.float32).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std, dtype=torch.float32).view(-1, 1, 1), False)
        self.num_frames = num_frames
        self.test_topk_per_image = test_topk_per_image
        self.apply_cls_thres = apply_cls_thres
        self.box_nms_thresh = box_nms_thresh
        # windowing (reference __init__ :150-155)
        self.num_frames_window_test = max(num_frames_window_test, num_frames)
        self.num_frames_window_output = (math.ceil(self.num_frames_window_test / 5) + 1) * 5
        self.clip_stride = clip_stride
        self.use_quasi_track = True
        self.temporal_consistency_threshold = temporal_consistency_threshold
        self.detect_newly_object_threshold = detect_newly_object_threshold
        self.detect_newly_interval_frames = detect_newly_interval_frames
        self.num_prev_frames_memory = num_prev_frames_memory
        self.custom_videos_enable = custom_videos_enable
        self.overlap_threshold = overlap_threshold
        # category ids (1-based, as the reference's metadata.thing_dataset_id_to_contiguous_id keys) that are "things"
        self.thing_dataset_ids = frozenset(int(c) for c in thing_dataset_ids)
        self.video_unified_inference_entities = video_unified_inference_entities
        # see inference_video: overlapping clips of one window share the pixel decoder's per-frame outputs (False: once per clip, as the
        # reference calls it)
        self.pixel_decoder_once_per_window = True
        self.frame_shard = None
        # {dataset name: (num_classes, start index)} slices of the class-embedding table; None = no slicing
        self.dataset_category_info = COMBINED_DATASETS_CATEGORY_INFO if dataset_category_info is None else dataset_category_info

    @classmethod
    def from_config(cls, cfg, dataset_category_info=None):
        test = cfg.MODEL.UniVS.TEST
        return {
            "hidden_dim": cfg.MODEL.MASK_FORMER.HIDDEN_DIM,
            "num_queries": cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES,
            "overlap_threshold_entity": cfg.MODEL.MASK_FORMER.TEST.OVERLAP_THRESHOLD_ENTITY,
            "stability_score_thresh": cfg.MODEL.MASK_FORMER.TEST.STABILITY_SCORE_THRESH,
            "size_divisibility": cfg.MODEL.MASK_FORMER.SIZE_DIVISIBILITY,
            "pixel_mean": cfg.MODEL.PIXEL_MEAN,
            "pixel_std": cfg.MODEL.PIXEL_STD,
            "num_frames": cfg.INPUT.SAMPLING_FRAME_NUM,
            "test_topk_per_image": cfg.TEST.DETECTIONS_PER_IMAGE,
            "apply_cls_thres": cfg.MODEL.BoxVIS.TEST.APPLY_CLS_THRES,
            "box_nms_thresh": test.BOX_NMS_THRESH,
            "num_frames_window_test": cfg.MODEL.BoxVIS.TEST.NUM_FRAMES_WINDOW,
            "clip_stride": cfg.MODEL.BoxVIS.TEST.CLIP_STRIDE,
            "num_prev_frames_memory": test.NUM_PREV_FRAMES_MEMORY,
            "video_unified_inference_entities": test.VIDEO_UNIFIED_INFERENCE_ENTITIES,
            "temporal_consistency_threshold": test.TEMPORAL_CONSISTENCY_THRESHOLD,
            "detect_newly_object_threshold": test.DETECT_NEWLY_OBJECT_THRESHOLD,
            "detect_newly_interval_frames": test.DETECT_NEWLY_INTERVAL_FRAMES,
            "custom_videos_enable": test.CUSTOM_VIDEOS_ENABLE,
            "dataset_category_info": dataset_category_info,
            "overlap_threshold": cfg.MODEL.MASK_FORMER.TEST.OVERLAP_THRESHOLD,
        }

    @property
    def device(self):
        return self.pixel_mean.device

    # ------------------------------------------------------------------------------------------
    # entry points
    # ------------------------------------------------------------------------------------------
    def eval(self, model, batched_inputs, targets=None):
        """The reference's entry point (:237-281): batched_inputs = [{"image": [frame tensors CHW, 0..255], "video_len",
        "height", "width", "task", "dataset_name", "file_names", ...}].  `targets` defaults to what
        `model.prepare_targets.process_inference` builds (the reference's only mode); a prepared list may be passed."""
        frames = [f.to(self.device) for video in batched_inputs for f in video["image"]]
        images = normalized_image_list(frames, self.pixel_mean, self.pixel_std, self.size_divisibility)
        if targets is None:
            targets = model.prepare_targets.process_inference(batched_inputs, tuple(images.tensor.shape[-2:]), self.device,
                                                              getattr(model, "text_prompt_encoder", None),
                                                              images.image_sizes[0])
        if self.video_unified_inference_entities:
            targets[0]["sub_task"] = self.video_unified_inference_entities
        else:
            name = targets[0]["dataset_name"]
            if name.startswith("ytvis") or name.startswith("ovis"):
                targets[0]["sub_task"] = "vis"
            elif name.startswith("vipseg"):
                targets[0]["sub_task"] = "vps"
            elif name.startswith("vspw"):
                targets[0]["sub_task"] = "vss"
            else:
                raise ValueError(f"Not support to eval the dataset {name} yet")
        return self.inference_video(model, batched_inputs, images, targets)

    def _class_slice(self, sub_task, dataset_name):
        if sub_task.startswith("entity"):
            if sub_task not in ENTITY_SUBTASK_DATASET:
                raise ValueError(sub_task)
            dataset_name = ENTITY_SUBTASK_DATASET[sub_task]
            return self.dataset_category_info[dataset_name]
        return self.dataset_category_info.get(dataset_name)

    def set_frame_shard(self, shard):
        """Run `inference_video` with the frames of the video spread over the ranks of `shard` (a univs_amd.distributed.FrameShard:
        group, rank, world; None switches back).  Frame f belongs to rank f % world: a window's backbone + pixel decoder run on the
        rank's own frames only, once per frame AND WINDOW: with `num_frames_window_test` > `num_frames` a frame serves every clip of its
        window (the reference's loop runs the pixel decoder once per CLIP: five times per frame at stride 1); with the shipped
        configs' window == clip length every clip opens a new window and its frames are recomputed -- and every clip's decoder runs on all ranks, each with the
        clip's frames it owns (ClipShard: one all-gather of the query states per layer).  The per-video state `targets[0]` stays
        REPLICATED: after a clip, the mask logits / embeddings of its frames are all-gathered and every rank does the same
        book-keeping.  Needs world <= num_frames (every rank must own a frame of every clip; on a larger node the other ranks take
        other videos).  SURVEY.md 8e; reference loop: inference_video_entity.py:296-316."""
        self.frame_shard = shard

    def inference_video(self, model, batched_inputs, images, targets, merge_results=True, on_clip=None):
        """`on_clip(first_frame_idx, targets)`: optional observer called at the entry of every clip (tests, tracing)."""
        x = images.tensor
        tv = targets[0]
        sub_task = tv["sub_task"]
        if not any(t in sub_task for t in ("vis", "vss", "vps")):
            raise ValueError(f"Not support to eval the sub-task {sub_task!r} yet")
        n_total = len(x)
        video_len = int(batched_inputs[0]["video_len"])
        interim_size = tuple(x.shape[-2:])
        image_size = tuple(images.image_sizes[0])
        out_size = (batched_inputs[0].get("height", image_size[0]), batched_inputs[0].get("width", image_size[1]))
        T = self.num_frames
        stride = min(T if "vss" in sub_task else self.clip_stride, T)   # semantic: non-overlapping clips (:299-300)

        results = []
        is_last = False
        win_start = win_end = 0
        feats_window = None
        shard = check_loop_shard(getattr(self, "frame_shard", None), T)
        # instance sub-task in a sharded loop: the clip's mask logits stay where they were computed (ClipMaskRows)
        lazy_ok = shard is not None and "vis" in sub_task and self.use_quasi_track and not getattr(self, "replicate_clip_masks", False)
        lazy = None
        begin_video(model, x.device, shard)
        win_rows, win_pd = {}, None
        for i in range(0, n_total, stride):
            if is_last and i + T > n_total:
                break
            is_last = i + T >= n_total
            tv["first_frame_idx"] = i
            tv["frame_indices"] = torch.arange(i, min(i + T, n_total))

            if on_clip is not None:
                on_clip(i, targets)
            if shard is not None:
                if i + T > win_end:  # the window's frames: backbone AND pixel decoder, on their owners, once per frame
                    win_start, win_end = i, i + self.num_frames_window_test
                    win_rows, win_pd = window_features_on_owner(model, x, list(range(win_start, min(win_end, n_total))), shard)
                out = sharded_clip_forward(model, targets, i, min(T, n_total - i), win_rows, win_pd, shard, device=x.device,
                                           lazy_masks=lazy_ok)
                if lazy_ok:
                    out, lazy = out
            else:
                if i + T > win_end:      # the backbone runs once per window of frames
                    win_start, win_end = i, i + self.num_frames_window_test
                    feats_window = model.backbone(x[win_start:win_end])
                    # windows longer than a clip: the pixel decoder too runs once per FRAME of the window (it is per-frame work; the
                    # reference's loop repeats it for every clip a frame belongs to -- five times per frame at stride 1, :316)
                    win_pd = None
                    head = model.sem_seg_head
                    if (self.pixel_decoder_once_per_window and self.num_frames_window_test > T and stride < T
                            and hasattr(head, "pixel_decoder") and hasattr(head, "predictor")):
                        mf, bfe, _enc, ms = head.pixel_decoder.forward_features(feats_window)
                        win_pd = (mf, bfe, list(ms))
                o = i - win_start
                if win_pd is not None:
                    mf, bfe, ms = win_pd
                    out = model.sem_seg_head.predictor([lv[o:o + T] for lv in ms], mf[o:o + T], bfe[o:o + T] if bfe is not None else None,
                                                       None, targets)
                else:
                    feats = {k: v[o:o + T] for k, v in feats_window.items()}
                    out = model.sem_seg_head(feats, targets=targets)
            out.pop("aux_outputs", None)

            out["pred_logits"] = out["pred_logits"].sigmoid()
            sl = self._class_slice(sub_task, tv["dataset_name"])
            if sl is not None:
                n_cls, start = sl
                assert start + n_cls <= out["pred_logits"].shape[-1]
                out["pred_logits"] = out["pred_logits"][..., start:start + n_cls]

            out = {k: v[0] for k, v in out.items() if v is not None}       # batch of one video
            out = {k: v for k, v in out.items() if v is not None}          # e.g. pred_reid_logits = [None]
            out_learn = {k: v[: self.num_queries] for k, v in out.items()}
            out_prompt = {k: v[self.num_queries:] for k, v in out.items()}
            if lazy is not None:
                # the prompt queries' logits enter the replicated state whole: gathered (few rows); the learnable queries' stay sharded
                out_prompt["pred_masks"] = lazy.rows(torch.arange(self.num_queries, lazy.n_rows))
                out_learn.pop("pred_masks", None)

            if "vss" in sub_task:
                # semantic segmentation needs no entity bookkeeping: classes x masks of the learnable queries
                results.append(self.save_results_vss(i, out_learn, interim_size, image_size, out_size, is_last, stride))
                continue
            # 1. entities already in the pool: accumulate this clip's predictions of their prompt queries
            self.write_prompt_predictions_into_annotations_per_clip(i, out_prompt, targets, interim_size, image_size, stride)
            # 2. new entities from the learnable queries
            if i % self.detect_newly_interval_frames == 0 or tv["masks"].nelement() == 0:
                if "vis" in sub_task:
                    self.detect_newly_entities_per_clip_instance(out_learn, targets, interim_size, sharded=lazy)
                else:
                    self.detect_newly_entities_per_clip_pixel(out_learn, targets, interim_size)
                self.write_newly_entities_into_annotations_per_clip(i, out_learn, targets, interim_size)
            # 3. emit finished frames
            is_out = i > self.num_prev_frames_memory and i % self.num_frames_window_output == self.num_prev_frames_memory
            if is_out or is_last:
                save = self.save_results_vis if "vis" in sub_task else self.save_results_vps
                results.append(save(i, targets, interim_size, image_size, out_size, is_last))
                w = self.num_frames_window_output
                tv["mask_logits"] = tv["mask_logits"][:, w:]
                tv["masks"] = tv["masks"][:, w:]
                tv["occurrence"] = tv["occurrence"][:, w:]
            # 4. room for the next clip's new frames
            if not is_last and "masks" in tv:
                self.pad_zero_annotations_for_next_clip(targets, min(stride, video_len - i - T))
        if "vss" in sub_task:
            return self.vss_output_results(targets, results, out_size)
        if "vps" in sub_task:
            return self.vps_output_results(targets, results, out_size)
        if not merge_results:
            return results            # per clip: [{"obj_id", "score", "masks" bool [T', H, W], "frame_id_start", ...}]
        # the reference's return value (:405-409): one YouTube-VIS record per (entity, class), COCO-RLE segmentations
        from .results import vis_clip_instances_to_coco_json_video
        if not ("video_id" in batched_inputs[0] and "height" in batched_inputs[0] and "width" in batched_inputs[0]):
            batched_inputs = [dict(batched_inputs[0], video_id=batched_inputs[0].get("video_id", 0), height=out_size[0],
                                   width=out_size[1])]
        return vis_clip_instances_to_coco_json_video(batched_inputs, results, test_topk_per_video=self.test_topk_per_image)

    # ------------------------------------------------------------------------------------------
    # step 1: prompt-specified entities
    # ------------------------------------------------------------------------------------------
    def write_prompt_predictions_into_annotations_per_clip(self, first_frame_idx, out, targets, interim_size, image_size, stride):
        if out["pred_masks"].nelement() == 0:
            return                                            # no prompt queries on this clip
        tv = targets[0]
        pred_embds = out["pred_embds"]                        # [N, T, C]
        pred_masks = _resize(out["pred_masks"], interim_size)  # [N, T, H, W] logits
        T = pred_masks.shape[1]

        thr = self.temporal_consistency_threshold * (0.5 if first_frame_idx < self.num_frames else 1.0)
        n_prev = max(int(self.num_prev_frames_memory / stride), 3)
        keep, sim = check_consistency_with_prev_frames(tv["embds"][:, -n_prev:], pred_embds, sim_threshold=thr,
                                                       return_similarity=True)

        cur = pred_masks[:, :, : image_size[0], : image_size[1]]
        q_hi, q_lo, _ = _quality_counts_and_boxes(pred_masks, valid=image_size, boxes=False)
        quality = q_hi / q_lo                                 # calculate_mask_quality_scores(cur)
        if "vis" in tv["sub_task"]:
            # every pixel goes to the entity with the highest score x probability; an entity survives
            # if it keeps enough of its own area
            score = tv["logits"].mean(1).max(-1)[0] * sim * quality
            prob = cur.sigmoid().flatten(1)                   # [N, T*h*w]
            fg = prob > 0.5
            owner = (score.view(-1, 1) * prob).argmax(0)
            owner = torch.where((prob < 0.5).all(0), torch.full_like(owner, -1), owner)   # background pixels
            own = owner[None] == torch.arange(len(prob), device=prob.device).view(-1, 1)
            # pixel counts per entity in two stages (image rows first): ATen reduces a [N, 4.6 M] bool tensor along its long axis with a
            # few workgroups per row -- 1.4 ms per sum at ten entities, 4.3 of the 5.9 ms this step took (tools/bench_write_prompt_pieces.py)
            wlast = cur.shape[-1]

            def count(b):
                return count_true(b.view(b.shape[0], -1, wlast))
            ratio = count(own) / count(fg).clamp(min=1)
            keep = keep & (ratio > self.overlap_threshold_entity) & (count(own & fg) > 0)

        idx = keep.nonzero(as_tuple=True)[0]                   # (one host round trip: the count tells whether anything is kept)
        if idx.numel():
            # The reference's lines (:466-475) on index sets -- `x[idx, -T:] += m` etc. -- each copy the kept entities' full-resolution
            # logits (188 MB at ten entities) several times; here the same values with the history updated in place: the kept rows are
            # added into the view of the last T frames, occurrence and boxes come from per-plane statistics read in place
            norm = _norm4(interim_size[1], interim_size[0], pred_masks.device)
            all_kept = idx.numel() == pred_masks.shape[0]
            recent = tv["mask_logits"][:, -T:]                # a view: [N, T, H, W]
            tv["occurrence"][idx, -T:] += _plane_stats(pred_masks)[..., 6].float()[idx]      # m.flatten(-2).gt(0).any(-1)
            if all_kept:
                recent += pred_masks
            else:
                recent.index_add_(0, idx, pred_masks.index_select(0, idx))
            tv["boxes"][idx, -T:] = (_plane_stats(recent)[..., 2:6].long()[idx] / norm.view(1, 1, -1)).to(tv["boxes"].dtype)
            last = tv["embds"][idx, -1]
            nonblank = (last != 0).any(-1)
            tv["embds"][idx, -1] = (last + pred_embds[idx].mean(1)) / (nonblank[..., None] + 1.0)
            tv["mask_quality_scores"][idx] += quality[idx]
        _refresh_recent_masks(tv, T)

    # ------------------------------------------------------------------------------------------
    # step 2: new entities
    # ------------------------------------------------------------------------------------------
    def detect_newly_entities_per_clip_instance(self, out_learn, targets, interim_size, sharded=None):
        """New entities among the learnable queries of a clip (inference_video_entity.py:560-652).  Same decisions and the same values
        as `_detect_newly_entities_per_clip_instance_direct`, arranged for a device: the per-pixel reductions (quality counts, boxes)
        run once for all queries, ONE device-to-host copy brings their results (exact small integers) and the class scores over, the
        stability filter / score order / box NMS / thresholds are evaluated on the host on [Q]-sized tensors -- the same IEEE
        operations on the same numbers --, and the mask tensor is touched by index only: rows matched to known entities and rows that
        become new entities are gathered once each.  Host round trips per clip: 1 on a video's first clip, 3 later (the assignment
        problem is solved on the host in the reference too), against ~10 boolean-index round trips before."""
        if not self.use_quasi_track:       # (never set by the reference's configs: its other matcher stays on the direct form)
            return self._detect_newly_entities_per_clip_instance_direct(out_learn, targets, interim_size)
        tv = targets[0]
        logits = out_learn["pred_logits"].float()     # [Q, K] probabilities
        embds = out_learn["pred_embds"].float()       # [Q, T, C]
        if sharded is None:
            masks = out_learn["pred_masks"].float()   # [Q, T, h, w]
            dev = masks.device
            Q, T = masks.shape[:2]
            h, w = masks.shape[-2:]
            take = lambda rows_: masks[rows_]                                                  # noqa: E731
        else:
            # frame-sharded loop (`sharded`: ClipMaskRows): the logits stay on the ranks of their frames; the learnable queries are its
            # first rows.  Statistics, the few rows that enter the per-video state and the candidates' IoU are collectives
            masks, dev = None, logits.device
            Q, T = logits.shape[0], sharded.T
            h, w = sharded.hw
            take = sharded.rows
        first = "masks" not in tv

        # ---- device, all queries: |{logit > 1}|, |{logit > -1}| (calculate_mask_quality_scores), integer boxes
        hi, lo, boxes_i = _quality_counts_and_boxes(masks, stats=None if sharded is None else sharded.stats()[:Q])   # [Q], [Q], [Q, T, 4]
        quality_d = hi / lo
        logits_d = logits * quality_d.view(-1, 1)
        norm_d = _norm4(w, h, dev)
        # ---- one copy to the host (counts and box corners are integers below 2^24: exact in float32).  The host side is NUMPY on
        # purpose: a torch CPU operator on ~10^5 elements opens an OpenMP region, and on a 256-thread host the pool's spin-waiting
        # afterwards slowed the whole clip loop four-fold (measured: 186 -> 760 ms per video); numpy's float32 arithmetic is the
        # same IEEE operation per element
        K = logits.shape[-1]
        host = torch.cat([hi.float()[:, None], lo.float()[:, None], boxes_i.flatten(1).float(), logits], 1).cpu().numpy()
        quality = host[:, 0] / host[:, 1]
        boxes = host[:, 2:2 + 4 * T].reshape(Q, T, 4) / np.asarray([w, h, w, h], dtype=np.float32)
        lg = host[:, 2 + 4 * T:2 + 4 * T + K] * quality[:, None]
        idx = np.arange(Q)                                                                      # surviving rows -> query index
        if self.stability_score_thresh > 0.0:
            k = quality > np.float32(self.stability_score_thresh)
            lg, boxes, quality, idx = lg[k], boxes[k], quality[k], idx[k]
        scores = lg.max(-1) if len(idx) else np.zeros((0,), np.float32)
        top = np.argsort(-scores, kind="stable")[: self.test_topk_per_image]
        lg, boxes, quality, idx, scores = lg[top], boxes[top], quality[top], idx[top], scores[top]
        if len(idx) > 1:
            order = np.argsort(-scores, kind="stable")
            bo = boxes[order]
            area = (bo[..., 2] - bo[..., 0]) * (bo[..., 3] - bo[..., 1])                         # utils/comm.py: video_box_iou, same order
            lt = np.maximum(bo[:, None, :, :2], bo[None, :, :, :2])
            rb = np.minimum(bo[:, None, :, 2:], bo[None, :, :, 2:])
            wh = np.maximum(rb - lt, np.float32(0))
            inter = wh[..., 0] * wh[..., 1]
            union = np.maximum(area[:, None] + area[None] - inter, np.float32(1e-3))
            biou = (inter / union).max(-1)
            worst = np.triu(biou, k=1).max(0)
            k = order[worst < np.float32(self.box_nms_thresh)]
            lg, boxes, quality, idx = lg[k], boxes[k], quality[k], idx[k]
        n = len(idx)
        best = lg.max(-1) if n else np.zeros((0,), np.float32)
        idx = torch.from_numpy(np.ascontiguousarray(idx))

        if first:
            new = torch.from_numpy(best > np.float32(max(self.apply_cls_thres, 0.1)))
        else:
            idx_d = to_device_async(idx, dev)            # (pinned staging: a pageable copy would wait for the stream, hazard 14)
            embds_s = embds.index_select(0, idx_d)
            gt_embds = tv["embds"]
            tgt = gt_embds[:, -3:]
            sim = torch.einsum("ntc,mfc->nmtf", tgt, embds_s).flatten(2)
            sim = (sim.softmax(1) + sim.softmax(0)).mean(-1) / 2.0
            sim = torch.where(sim < self.detect_newly_object_threshold, torch.zeros_like(sim), sim).cpu().numpy()   # (round trip 2)
            rows, cols = linear_sum_assignment(np.float32(1) - sim)
            msim = sim[rows, cols]
            rows, cols = torch.from_numpy(rows.astype(np.int64)), torch.from_numpy(cols.astype(np.int64))

            ok = torch.from_numpy(msim > np.float32(self.detect_newly_object_threshold))
            r, c = to_device_async(rows[ok], dev), to_device_async(cols[ok], dev)
            # the stored class scores / embeddings follow the learnable queries (prompt and learnable
            # queries live in slightly different feature spaces)
            tv["logits"][r, -1] = 0.5 * (tv["logits"][r, -1] + logits_d[idx_d[c]])
            last = gt_embds[r, -1]
            gt_embds[r, -1] = (last + embds_s[c].mean(1)) / ((last != 0).any(-1)[..., None] + 1.0)

            ok2 = torch.from_numpy(msim > np.float32(2 * self.detect_newly_object_threshold))
            c2_h = cols[ok2]
            r2, c2 = to_device_async(rows[ok2], dev), idx_d[to_device_async(c2_h, dev)]       # (c2: query indices)
            m = _resize(take(c2), interim_size)
            tv["occurrence"][r2, -T:] += _plane_stats(m)[..., 6].float()                          # m.flatten(-2).gt(0).any(-1)
            tv["mask_logits"][:, -T:].index_add_(0, r2, m)       # x[r2, -T:] += m without the copy out and back (r2: distinct rows)
            tv["mask_quality_scores"][r2] += quality_d[c2]
            _refresh_recent_masks(tv, T)

            # a query is a NEW entity if it is unmatched, confident, and overlaps no known entity
            # (mask IoU < 0.5 in every frame of the clip)
            cand = torch.ones(n, dtype=torch.bool)
            cand[c2_h] = False     # (the reference excludes the strongly matched queries here, :640-642)
            cand &= torch.from_numpy(best > np.float32(self.apply_cls_thres))
            n_known = tv["mask_logits"].shape[0]
            if n_known > 0 and n > 0:
                ci = cand.nonzero(as_tuple=True)[0]
                if len(ci):        # (the IoU only of the rows that can still qualify: the host knows them)
                    known = _resize(tv["mask_logits"][:, -T:], (h, w)).transpose(0, 1).gt(0.0)          # [T, N, h, w]
                    rows_c = idx_d[to_device_async(ci, dev)]
                    if sharded is None:
                        worst_iou = batched_mask_iou(masks[rows_c].transpose(0, 1).gt(0.0), known).amax(dim=(0, 2))    # [T, n_c, N] -> [n_c]
                    else:
                        worst_iou = sharded.iou_max(rows_c, known)
                    cand[ci] = (worst_iou < 0.5).cpu()
            else:
                cand &= False          # reference: an empty IoU matrix never qualifies (:644)
            new = cand

        sel = to_device_async(idx[new], dev)
        out_learn["pred_logits"] = logits_d[sel]
        out_learn["pred_masks"] = take(sel)
        out_learn["pred_embds"] = embds[sel]
        out_learn["pred_boxes"] = boxes_i[sel] / norm_d
        out_learn["mask_quality_scores"] = quality_d[sel]

    def _detect_newly_entities_per_clip_instance_direct(self, out_learn, targets, interim_size):
        """The reference's formulation line by line (inference_video_entity.py:560-652): every filter is a boolean index of the mask
        tensor (a host round trip and a copy of up to [Q, T, h, w] each).  Kept as the oracle of `detect_newly_entities_per_clip_instance`
        (tests/test_clip_loop_cpu.py compares the two on random scenes)."""
        tv = targets[0]
        logits = out_learn["pred_logits"].float()     # [Q, K] probabilities
        masks = out_learn["pred_masks"].float()       # [Q, T, h, w]
        embds = out_learn["pred_embds"].float()       # [Q, T, C]
        T = masks.shape[1]
        first = "masks" not in tv

        quality = calculate_mask_quality_scores(masks)
        logits = logits * quality.view(-1, 1)
        if self.stability_score_thresh > 0.0:
            k = quality > self.stability_score_thresh
            logits, masks, embds, quality = logits[k], masks[k], embds[k], quality[k]

        scores = logits.max(-1)[0]
        top = scores.sort(descending=True)[1][: self.test_topk_per_image]
        logits, masks, embds, quality, scores = logits[top], masks[top], embds[top], quality[top], scores[top]

        h, w = masks.shape[-2:]
        boxes = convert_mask_to_box(masks > 0) / _norm4(w, h, masks.device)
        if masks.shape[0] > 1:
            # box-IoU NMS over the clip: drop a query whose boxes overlap a better one's in any frame
            order = scores.sort(descending=True)[1]
            biou = video_box_iou(boxes[order], boxes[order])[0].max(-1)[0]
            worst = torch.triu(biou, diagonal=1).max(0)[0]
            k = order[worst < self.box_nms_thresh]
            logits, masks, embds, boxes, quality = logits[k], masks[k], embds[k], boxes[k], quality[k]

        if first:
            new = logits.max(-1)[0] > max(self.apply_cls_thres, 0.1)
        else:
            gt_embds = tv["embds"]
            tgt = gt_embds[:, -3:]
            if self.use_quasi_track:
                sim = torch.einsum("ntc,mfc->nmtf", tgt, embds).flatten(2)
                sim = (sim.softmax(1) + sim.softmax(0)).mean(-1) / 2.0
                sim = torch.where(sim < self.detect_newly_object_threshold, torch.zeros_like(sim), sim)
                rows, cols = linear_sum_assignment((1 - sim).cpu())
            else:
                (rows, cols), _ = match_from_learnable_embds(tgt, embds, return_similarity=True, return_src_indices=True,
                                                             use_norm=True, thresh=self.detect_newly_object_threshold)
            rows = torch.as_tensor(rows, device=sim.device)
            cols = torch.as_tensor(cols, device=sim.device)
            msim = sim[rows, cols]

            ok = msim > self.detect_newly_object_threshold
            r, c = rows[ok], cols[ok]
            # the stored class scores / embeddings follow the learnable queries (prompt and learnable
            # queries live in slightly different feature spaces)
            tv["logits"][r, -1] = 0.5 * (tv["logits"][r, -1] + logits[c])
            last = gt_embds[r, -1]
            gt_embds[r, -1] = (last + embds[c].mean(1)) / ((last != 0).any(-1)[..., None] + 1.0)

            ok2 = msim > 2 * self.detect_newly_object_threshold
            r2, c2 = rows[ok2], cols[ok2]
            m = _resize(masks[c2], interim_size)
            tv["occurrence"][r2, -T:] += m.flatten(-2).gt(0.0).any(-1).float()
            tv["mask_logits"][r2, -T:] += m
            tv["mask_quality_scores"][r2] += quality[c2]
            tv["masks"] = tv["mask_logits"].gt(0.0).float()

            # a query is a NEW entity if it is unmatched, confident, and overlaps no known entity
            # (mask IoU < 0.5 in every frame of the clip)
            known = _resize(tv["mask_logits"][:, -T:], masks.shape[-2:]).transpose(0, 1).gt(0.0)   # [T, N, h, w]
            cand = torch.ones(len(masks), dtype=torch.bool, device=masks.device)
            cand[c2] = False     # (the reference excludes the strongly matched queries here, :640-642)
            cand &= logits.max(-1)[0] > self.apply_cls_thres
            if known.shape[1] > 0 and len(masks) > 0:
                miou = batched_mask_iou(masks.transpose(0, 1).gt(0.0), known)      # [T, Q, N]
                cand &= miou.amax(dim=(0, 2)) < 0.5
            else:
                cand &= False          # reference: an empty IoU matrix never qualifies (:644)
            new = cand

        for k_, v in (("pred_logits", logits), ("pred_masks", masks), ("pred_embds", embds), ("pred_boxes", boxes),
                      ("mask_quality_scores", quality)):
            out_learn[k_] = v[new]

    def detect_newly_entities_per_clip_pixel(self, out_learn, targets, interim_size):
        """Panoptic variant (inference_video_entity.py:654-765): things are de-duplicated by box IoU, stuff by the
        mask IoU of the first frame; matched entities always absorb the learnable queries' masks."""
        tv = targets[0]
        first = "masks" not in tv
        logits = out_learn["pred_logits"].float()
        masks = out_learn["pred_masks"].float()
        embds = out_learn["pred_embds"].float()
        h, w = masks.shape[-2:]
        T = masks.shape[1]
        boxes = convert_mask_to_box(masks > 0) / _norm4(w, h, masks.device)
        quality = calculate_mask_quality_scores(masks)
        logits = logits * quality.view(-1, 1)
        scores, labels = logits.max(-1)

        if first:
            order = scores.sort(descending=True)[1][:100]
            isthing = torch.as_tensor([int(c) + 1 in self.thing_dataset_ids for c in labels[order].tolist()],
                                      dtype=torch.bool, device=order.device)
            things, stuff = order[isthing], order[~isthing]
            if len(things):
                things = things[:70]
                biou = video_box_iou(boxes[things], boxes[things])[0].max(-1)[0]
                things = things[torch.triu(biou, diagonal=1).max(0)[0] < self.box_nms_thresh]
            if len(stuff):
                stuff = stuff[:30]
                m0 = masks[stuff][:, 0].gt(0.0).float().unsqueeze(0)
                miou = batched_mask_iou(m0, m0).max(0)[0]
                stuff = stuff[torch.triu(miou, diagonal=1).max(0)[0] < 0.6]
            new = torch.cat([things, stuff])
            new = new[scores[new] > self.apply_cls_thres]
        else:
            gt_embds = tv["embds"]
            tgt = gt_embds[:, -3:]
            if self.use_quasi_track:
                sim = torch.einsum("ntc,mfc->nmtf", tgt, embds).flatten(2)
                sim = (sim.softmax(1) + sim.softmax(0)).mean(-1) / 2.0
                sim = torch.where(sim < self.detect_newly_object_threshold, torch.zeros_like(sim), sim)
                rows, cols = linear_sum_assignment((1 - sim).cpu())
                rows = torch.as_tensor(rows, device=sim.device)
                cols = torch.as_tensor(cols, device=sim.device)
                msim = sim[rows, cols]
            else:
                (rows, cols), msim = match_from_learnable_embds(tgt, embds, return_similarity=True, return_src_indices=True,
                                                                use_norm=False, thresh=self.detect_newly_object_threshold)
                rows = torch.as_tensor(rows, device=msim.device)
                cols = torch.as_tensor(cols, device=msim.device)
            ok = msim > self.detect_newly_object_threshold
            r, c = rows[ok], cols[ok]
            m = _resize(masks[c], interim_size)
            tv["mask_logits"][r, -T:] += m
            tv["occurrence"][r, -T:] += m.flatten(-2).gt(0.0).any(-1).float()
            tv["logits"][r, -1] = 0.5 * (tv["logits"][r, -1] + logits[c])
            last = gt_embds[r, -1]
            gt_embds[r, -1] = (last + embds[c].mean(1)) / ((last != 0).any(-1)[..., None] + 1.0)
            tv["mask_quality_scores"][r] += quality[c]
            tv["masks"] = tv["mask_logits"].gt(0.0).float()

            known = _resize(tv["mask_logits"][:, -T:], masks.shape[-2:]).transpose(0, 1).gt(0.0)   # [T, N, h, w]
            cand = torch.ones(len(masks), dtype=torch.bool, device=masks.device)
            cand[c] = False
            cand &= scores > 2 * self.apply_cls_thres
            if known.shape[1] > 0 and len(masks) > 0:
                cand &= batched_mask_iou(masks.transpose(0, 1).gt(0.0), known).amax(dim=(0, 2)) < 0.5
            else:
                cand &= False
            new = cand.nonzero(as_tuple=True)[0]

        for k_, v in (("pred_logits", logits), ("pred_masks", masks), ("pred_embds", embds), ("pred_boxes", boxes),
                      ("mask_quality_scores", quality)):
            out_learn[k_] = v[new]

    def write_newly_entities_into_annotations_per_clip(self, first_frame_idx, out, targets, interim_size):
        tv = targets[0]
        dev = out["pred_masks"].device
        logits = out["pred_logits"].unsqueeze(1)                       # [n, 1, K]
        embds = out["pred_embds"].mean(dim=1, keepdim=True)            # [n, 1, C]
        boxes = out["pred_boxes"]                                      # [n, T, 4]
        quality = out["mask_quality_scores"]
        n, T = out["pred_masks"].shape[:2]
        masks = _resize(out["pred_masks"], interim_size) if n else torch.zeros((0, self.num_frames) + tuple(interim_size), device=dev)
        occ = torch.ones(masks.shape[:2], device=dev)
        first_idx = torch.full((n,), first_frame_idx, dtype=torch.long, device=dev)

        if "masks" not in tv:
            tv.update({"logits": logits, "masks": masks.gt(0.0), "mask_logits": masks, "boxes": boxes, "embds": embds,
                       "ids": torch.arange(n, device=dev), "first_appear_frame_idxs": first_idx,
                       "mask_quality_scores": quality, "occurrence": occ})
            return
        if n == 0:
            return

        def left_pad(new, old):          # zero history in front so that the new rows line up with `old` in time
            shape = list(old.shape)
            shape[0], shape[1] = n, old.shape[1] - new.shape[1]
            return torch.cat([new.new_zeros(shape), new], dim=1)

        n_old = len(tv["ids"])
        new_logits = left_pad(logits, tv["logits"])
        new_masks = left_pad(masks, tv["mask_logits"])
        tv.update({
            "logits": torch.cat([tv["logits"], new_logits]),
            "masks": torch.cat([tv["masks"], new_masks.gt(0.0)]),
            "mask_logits": torch.cat([tv["mask_logits"], new_masks]),
            "boxes": torch.cat([tv["boxes"], left_pad(boxes.float(), tv["boxes"])]),
            "embds": torch.cat([tv["embds"], left_pad(embds, tv["embds"])]),
            "ids": torch.cat([tv["ids"], torch.arange(n, device=dev) + n_old]),
            "occurrence": torch.cat([tv["occurrence"], left_pad(occ, tv["occurrence"])]),
            "first_appear_frame_idxs": torch.cat([tv["first_appear_frame_idxs"], first_idx]),
            "mask_quality_scores": torch.cat([tv["mask_quality_scores"], quality]),
        })
        if "prompt_pe" in tv:
            # the memory pool is indexed by entity: empty rows for the newcomers
            pe, pf, pm = tv["prompt_pe"], tv["prompt_feats"], tv["prompt_attn_masks"]
            tv["prompt_pe"] = torch.cat([pe, pe.new_zeros((n,) + tuple(pe.shape[1:]))])
            tv["prompt_feats"] = torch.cat([pf, pf.new_zeros((n,) + tuple(pf.shape[1:]))])
            tv["prompt_attn_masks"] = torch.cat(
                [pm, torch.zeros(pm.shape[0], pm.shape[1], n, pm.shape[-1], dtype=torch.bool, device=pm.device)], dim=-2)

    # ------------------------------------------------------------------------------------------
    # step 4 / 3
    # ------------------------------------------------------------------------------------------
    def pad_zero_annotations_for_next_clip(self, targets, stride):
        tv = targets[0]
        n = tv["embds"].shape[0]
        dev = tv["embds"].device
        H, W = tv["masks"].shape[-2:]
        zero_masks = torch.zeros((n, stride, H, W), dtype=torch.float, device=dev)
        tv.update({
            "logits": torch.cat([tv["logits"], tv["logits"][:, -1:].clone()], dim=1),
            "masks": torch.cat([tv["masks"], zero_masks], dim=1),
            "mask_logits": torch.cat([tv["mask_logits"], zero_masks], dim=1),
            "boxes": torch.cat([tv["boxes"], torch.zeros((n, stride, 4), device=dev)], dim=1),
            "embds": torch.cat([tv["embds"], tv["embds"][:, -3:].mean(dim=1, keepdim=True)], dim=1),
            "occurrence": torch.cat([tv["occurrence"], torch.zeros((n, stride), device=dev)], dim=1),
        })

    def save_results_vps(self, first_frame_idx, targets, interim_size, image_size, out_size, is_last):
        """Panoptic map of the finished frames (inference_video_entity.py:962-1053): every pixel goes to the entity
        with the highest score x logit, things keep an id per entity, stuff one id per class (both remembered in
        `targets[0]` across calls).  The area tests run on the device for all entities at once; the segment-id
        bookkeeping (a few dict updates per entity) stays on the host as in the reference."""
        tv = targets[0]
        masks, ids = tv["mask_logits"], tv["ids"]
        if not is_last:
            masks = masks[:, : self.num_frames_window_output]
        masks = masks[:, :, : image_size[0], : image_size[1]]
        masks = _resize(masks.float(), out_size) if masks.numel() else masks.new_zeros(masks.shape[:2] + tuple(out_size))
        if "stuff_memory_list" not in tv:
            tv["thing_memory_list"], tv["stuff_memory_list"] = {}, {}
        things_mem, stuff_mem = tv["thing_memory_list"], tv["stuff_memory_list"]
        known_ids = list(things_mem.values()) + list(stuff_mem.values())

        scores, classes = tv["logits"].mean(1).max(-1)
        classes = (classes + 1).tolist()                       # category labels start from 1
        scores = scores * calculate_mask_quality_scores(masks)
        is_thing = [c in self.thing_dataset_ids for c in classes]
        demote = torch.tensor([0.75 if (k not in things_mem and not is_thing[k]) else 1.0 for k in range(len(classes))],
                              device=scores.device, dtype=scores.dtype)
        scores = scores * demote                               # things win ties against stuff
        n, t = masks.shape[:2]
        panoptic = torch.zeros((t, out_size[0], out_size[1]), dtype=torch.int32, device=masks.device)
        if n == 0:
            return panoptic.cpu()
        assert int(ids.min()) == 0 and int(ids.max()) == len(ids) - 1
        owner = (scores.view(-1, 1, 1, 1) * masks).argmax(0)   # [t, h, w]
        prob = masks.sigmoid()
        owner = torch.where((prob < 0.5).all(0), torch.full_like(owner, -1), owner)
        onehot = owner[None] == torch.arange(n, device=owner.device).view(-1, 1, 1, 1)
        fg = prob >= 0.5
        area = count_true(onehot).tolist()
        orig = count_true(fg).tolist()
        inter = count_true(onehot & fg).tolist()
        seg_of = [0] * n
        nxt = max(known_ids) + 1 if known_ids else 0
        for k in range(n):
            if not (area[k] > 0 and orig[k] > 0 and inter[k] > 0):
                continue
            obj = int(ids[k])
            thr = 0.5 * self.overlap_threshold if obj in things_mem else self.overlap_threshold
            if is_thing[k] and area[k] / orig[k] < thr:
                continue
            mem, key = (things_mem, obj) if is_thing[k] else (stuff_mem, classes[k])
            if key not in mem:
                mem[key] = nxt + 1
                nxt += 1
            seg_of[k] = mem[key]
        table = torch.tensor(seg_of + [0], dtype=torch.int32, device=masks.device)    # owner -1 -> 0
        keep = torch.gather(fg, 0, owner.clamp(min=0)[None])[0] & (owner >= 0)
        panoptic = torch.where(keep, table[owner], panoptic)
        return panoptic.cpu()

    def vps_output_results(self, targets, panoptic_seg_list, out_size):
        tv = targets[0]
        classes = (tv["logits"].mean(1).max(-1)[1] + 1).tolist()
        infos = [{"id": seg, "isthing": classes[obj] in self.thing_dataset_ids, "category_id": int(classes[obj])}
                 for obj, seg in tv["thing_memory_list"].items()]
        infos += [{"id": seg, "isthing": False, "category_id": int(cls_)} for cls_, seg in tv["stuff_memory_list"].items()]
        return {"image_size": out_size, "pred_masks": torch.cat(panoptic_seg_list, dim=0).cpu(), "segments_infos": infos,
                "task": "vps"}

    def save_results_vss(self, first_frame_idx, output, interim_size, image_size, out_size, is_last, stride):
        """Per-clip semantic map (inference_video_entity.py:1086-1113): quality-weighted class probabilities
        times mask probabilities, argmax over classes.  [T', H_out, W_out] int64 on the host."""
        logits, masks = output["pred_logits"], output["pred_masks"]
        if not is_last:
            masks = masks[:, :stride]
        masks = _resize(masks, interim_size)[:, :, : image_size[0], : image_size[1]]
        masks = retry_if_oom(F.interpolate)(masks.float(), size=out_size, mode="nearest")
        logits = logits * calculate_mask_quality_scores(masks).view(-1, 1)
        semseg = torch.einsum("qc,qthw->cthw", logits, masks.sigmoid())
        return semseg.argmax(0).cpu()

    def vss_output_results(self, targets, sem_mask_list, out_size):
        return {"image_size": out_size, "pred_masks": torch.cat(sem_mask_list, dim=0).cpu(), "task": "vss"}

    def save_results_vis(self, first_frame_idx, targets, interim_size, image_size, out_size, is_last):
        tv = targets[0]
        if "masks" not in tv:
            return []
        frame_id_start = min(first_frame_idx + self.num_frames, tv["video_len"]) - tv["mask_logits"].shape[1]
        masks, occ = tv["mask_logits"], tv["occurrence"]
        if not is_last:
            masks, occ = masks[:, : self.num_frames_window_output], occ[:, : self.num_frames_window_output]
        masks = masks / occ[..., None, None].clamp(min=1)
        masks = masks[:, :, : image_size[0], : image_size[1]]
        masks = (_resize(masks.float(), out_size) > 0.0).cpu() if masks.numel() else masks.new_zeros(masks.shape[:2] + tuple(out_size)).bool().cpu()
        scores = tv["logits"].mean(1).cpu()
        quality = tv["mask_quality_scores"]
        res = []
        for i, obj_id in enumerate(tv["ids"].tolist()):
            r = {"obj_id": int(obj_id), "score": scores[i], "masks": masks[i], "frame_id_start": frame_id_start}
            if is_last:
                r["mask_quality_score"] = quality[i] / (int(quality.max()) + 1)
            res.append(r)
        return res
