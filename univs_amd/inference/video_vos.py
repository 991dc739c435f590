"""Clip loop for prompt-specified segmentation in videos: VOS (task 'sot', masks given where an object first
appears) and referring VOS (task 'grounding', expressions), around the hot path.

Counterpart of the reference's `InferenceVideoVOS` (univs/inference/inference_video_vos.py): same method names and the
same mutations of the per-video `targets[0]` dictionary (`masks`, `mask_logits`, `boxes`, `embds`, `labels`,
`first_appear_frame_idxs`, ...) that the prompt-as-query decoder reads on the next clip.

    inference_video_vos                         :243-284   windowed backbone, clip schedule
    write_targets_into_annotations_per_clip     :533-621   annotations of newly visible objects, room for new frames
    write_predictions_into_annotations_per_clip :286-531   (1) objects that first appear in this clip, (2) objects
                                                           followed from earlier clips; query sources 'prompt',
                                                           'learn' and 'prompt+learn' / 'learn+prompt'
    save_vos_results / save_rvos_results        :623-705   returned as tensors (object-id map / per-expression
                                                           masks); the PNG files are a result format

Per-frame annotations are duck-typed like detectron2 `Instances` (`ori_ids`, `gt_boxes.tensor`, `gt_masks` (tensor or
`.tensor`), `gt_classes`, `image_size`, `len()`, `.to(device)`); `FrameAnnotations` below is a minimal stand-in.
Datasets whose name contains 'viposeg' (panoptic VOS) take the reference's extras: a semantic map from the learnable
queries (class scores of the VIPSeg slice x mask probabilities) lends its pixels to 'stuff' objects (:320-325, :404-412,
:502-506), and re-identification among learnable queries uses un-normalised similarities with threshold 0.5 (:444-445).
"""
from typing import Tuple

import torch
import torch.nn.functional as F
from torch import nn

from ..registry import configurable
from ..utils.memory import retry_if_oom
from ..utils.comm import (batched_pair_mask_iou, calculate_mask_quality_scores, convert_mask_to_box, count_true, video_box_iou)
from .comm import check_consistency_with_prev_frames, match_from_learnable_embds


class FrameAnnotations:
    """Annotations of one frame: the objects whose masks are given there (normally the frame an object first appears)."""

    class _Boxes:
        def __init__(self, tensor):
            self.tensor = tensor

    def __init__(self, image_size, ori_ids=(), gt_masks=None, gt_boxes=None, gt_classes=None):
        self.image_size = tuple(image_size)
        self.ori_ids = list(ori_ids)
        self.gt_masks = gt_masks if gt_masks is not None else torch.zeros((0,) + self.image_size)
        self.gt_boxes = FrameAnnotations._Boxes(gt_boxes if gt_boxes is not None else torch.zeros((0, 4)))
        self.gt_classes = gt_classes if gt_classes is not None else torch.zeros(len(self.ori_ids), dtype=torch.long)

    def __len__(self):
        return len(self.ori_ids)

    def to(self, device):
        return FrameAnnotations(self.image_size, self.ori_ids, self.gt_masks.to(device), self.gt_boxes.tensor.to(device),
                                self.gt_classes.to(device))


def _resize(masks, size):
    # (the reference wraps these resizes in retry_if_cuda_oom, inference_video_entity.py:933 / :978 / :1104: utils/memory.py)
    return retry_if_oom(F.interpolate)(masks, size, mode="bilinear", align_corners=False)


class InferenceVideoVOS(nn.Module):
    @configurable
    def __init__(
        self,
        *,
        hidden_dim: int,
        num_queries: int,
        size_divisibility: int,
        pixel_mean: Tuple[float],
        pixel_std: Tuple[float],
        num_frames: int,
        prompt_as_queries: bool = True,
        num_frames_window_test: int = 5,
        clip_stride: int = 1,
        video_unified_inference_queries: str = "prompt",
        num_prev_frames_memory: int = 5,
        stuff_dataset_ids=(),
        use_semseg_pvos: bool = True,
        dataset_category_info=None,
    ):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.num_queries = num_queries
        self.size_divisibility = size_divisibility
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean, dtype=torch.float32).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std, dtype=torch.float32).view(-1, 1, 1), False)
        self.num_frames = num_frames
        self.prompt_as_queries = prompt_as_queries
        self.num_frames_window_test = max(num_frames_window_test, num_frames)
        self.clip_stride = clip_stride
        if video_unified_inference_queries not in ("prompt", "learn", "prompt+learn", "learn+prompt"):
            raise ValueError(f"video_unified_inference_queries={video_unified_inference_queries!r}")
        self.video_unified_inference_queries = video_unified_inference_queries
        self.num_prev_frames_memory = max(num_prev_frames_memory, num_frames)
        # VIPOSeg: 1-based dataset ids of the 'stuff' categories (the reference's metadata.stuff_dataset_id_to_contiguous_id keys)
        self.stuff_dataset_ids = frozenset(int(c) for c in stuff_dataset_ids)
        self.use_semseg_pvos = use_semseg_pvos
        from .video_entity import COMBINED_DATASETS_CATEGORY_INFO
        self.dataset_category_info = COMBINED_DATASETS_CATEGORY_INFO if dataset_category_info is None else dataset_category_info

    @classmethod
    def from_config(cls, cfg):
        return {
            "hidden_dim": cfg.MODEL.MASK_FORMER.HIDDEN_DIM,
            "num_queries": cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES,
            "size_divisibility": cfg.MODEL.MASK_FORMER.SIZE_DIVISIBILITY,
            "pixel_mean": cfg.MODEL.PIXEL_MEAN,
            "pixel_std": cfg.MODEL.PIXEL_STD,
            "num_frames": cfg.INPUT.SAMPLING_FRAME_NUM,
            "prompt_as_queries": cfg.MODEL.UniVS.PROMPT_AS_QUERIES,
            "num_frames_window_test": cfg.MODEL.BoxVIS.TEST.NUM_FRAMES_WINDOW,
            "clip_stride": cfg.MODEL.BoxVIS.TEST.CLIP_STRIDE,
            "video_unified_inference_queries": cfg.MODEL.UniVS.TEST.VIDEO_UNIFIED_INFERENCE_QUERIES,
            "num_prev_frames_memory": cfg.MODEL.UniVS.TEST.NUM_PREV_FRAMES_MEMORY,
        }

    @property
    def device(self):
        return self.pixel_mean.device

    # ------------------------------------------------------------------------------------------
    def eval(self, model, batched_inputs, targets=None):
        """The reference's entry point (:203-241): normalise and pad the frames, build `targets` through
        `model.prepare_targets.process_inference` (unless a prepared list is passed), run the loop."""
        from .video_entity import normalized_image_list
        frames = [f.to(self.device) for video in batched_inputs for f in video["image"]]
        images = normalized_image_list(frames, self.pixel_mean, self.pixel_std, self.size_divisibility)
        image_size = images.image_sizes[0]
        out_size = (batched_inputs[0].get("height", image_size[0]), batched_inputs[0].get("width", image_size[1]))
        if targets is None:
            targets = model.prepare_targets.process_inference(batched_inputs, tuple(images.tensor.shape[-2:]), self.device,
                                                              getattr(model, "text_prompt_encoder", None))
        targets[0]["video_len"] = len(frames)
        return self.inference_video_vos(model, batched_inputs, images, targets, image_size, out_size)

    def set_frame_shard(self, shard):
        """The clip loop with the frames of the video spread over the ranks of `shard` (frame f on rank f % world; backbone + pixel decoder
        on the owned frames, every clip's decoder on all ranks through ClipShard, `targets[0]` replicated): see
        InferenceVideoEntity.set_frame_shard.  None switches back."""
        self.frame_shard = shard

    def inference_video_vos(self, model, batched_inputs, images, targets, image_size=None, out_size=None):
        from .video_entity import begin_video, check_loop_shard, sharded_clip_forward, window_features_on_owner
        x = images.tensor
        image_size = tuple(images.image_sizes[0])
        out_size = tuple(out_size) if out_size is not None else image_size
        video_len = len(x)
        tv = targets[0]
        T = self.num_frames
        stride = min(self.clip_stride, T)
        results = []
        is_last = False
        win_start = win_end = 0
        feats_window = None
        shard = check_loop_shard(getattr(self, "frame_shard", None), T)
        begin_video(model, x.device, shard)
        win_rows, win_pd = {}, None
        for i in range(0, video_len, stride):
            if is_last and i + T > video_len:
                break
            is_last = i + T >= video_len
            tv["frame_indices"] = torch.arange(i, min(i + T, video_len))
            if i + T > win_end:
                win_start, win_end = i, min(i + self.num_frames_window_test, video_len)
                if shard is not None:    # backbone AND pixel decoder on the frames this rank owns, once per frame
                    win_rows, win_pd = window_features_on_owner(model, x, list(range(win_start, win_end)), shard)
                else:
                    feats_window = model.backbone(x[win_start:win_end])
            # 1. annotations of objects that become visible in this clip; room for the clip's new frames
            self.write_targets_into_annotations_per_clip(targets, i, stride)
            # 2. the hot path
            if shard is not None:
                out = sharded_clip_forward(model, targets, i, min(T, video_len - i), win_rows, win_pd, shard, device=x.device)
            else:
                o = i - win_start
                feats = {k: v[o:o + T] for k, v in feats_window.items()}
                out = model.sem_seg_head(feats, targets=targets)
            out.pop("aux_outputs", None)
            # 3. predictions -> pseudo annotations (the prompts of the following frames)
            self.write_predictions_into_annotations_per_clip(out, image_size, targets, i, stride)
            if tv["task"] == "sot" or "davis" in tv["dataset_name"]:
                results.append(self.save_vos_results(i, targets, image_size, out_size, is_last, stride))
            elif tv["task"] == "grounding":
                results.append(self.save_rvos_results(i, targets, image_size, out_size, is_last, stride))
        return results

    # ------------------------------------------------------------------------------------------
    def write_targets_into_annotations_per_clip(self, targets, first_frame_idx, stride):
        for tv in targets:
            video_len = tv["video_len"]
            h_pad, w_pad = tv["inter_image_size"]
            dev = self.device
            norm = torch.as_tensor([w_pad, h_pad, w_pad, h_pad], dtype=torch.float32, device=dev).reshape(1, -1)
            if "ids" not in tv:      # first clip of the video
                if tv["task"] == "grounding":
                    n = len(tv["exp_obj_ids"])
                    tv["ids"] = [int(o) for o in tv["exp_obj_ids"]]
                    tv["first_appear_frame_idxs"] = torch.zeros(n, dtype=torch.long, device=dev)
                    tv["labels"] = torch.ones(n, dtype=torch.bool, device=dev) * -1
                else:
                    ids = list(set(sum([t.ori_ids for t in tv["instances"]], [])))
                    tv["ids"] = [g for g in ids if g != -1]
                    tv["first_appear_frame_idxs"] = torch.ones(len(tv["ids"]), dtype=torch.long, device=dev) * -1
                    tv["labels"] = torch.ones(len(tv["ids"]), dtype=torch.bool, device=dev) * -1
            tv["first_frame_idx"] = first_frame_idx
            T = min(self.num_frames, video_len - first_frame_idx)         # the last clip may be shorter
            n = len(tv["ids"])
            t_new = T if first_frame_idx == 0 else min(stride, video_len - first_frame_idx)
            ids, labels, first_seen = tv["ids"], tv["labels"], tv["first_appear_frame_idxs"]
            masks = torch.zeros([n, t_new, h_pad, w_pad], dtype=torch.float, device=dev)
            mask_logits = masks.clone()
            boxes = torch.zeros([n, t_new, 4], dtype=torch.float32, device=dev)
            if first_frame_idx == 0:
                embds = torch.zeros([n, t_new, self.hidden_dim], dtype=torch.float32, device=dev)
            else:
                embds = tv["embds"][:, -t_new:].mean(1).unsqueeze(1).repeat(1, t_new, 1).clone()
                keep = self.num_prev_frames_memory       # masks of the last frames only (memory), boxes / embds all
                masks = torch.cat([tv["masks"][:, -keep:], masks], dim=1)
                mask_logits = torch.cat([tv["mask_logits"][:, -keep:], mask_logits], dim=1)
                boxes = torch.cat([tv["boxes"], boxes], dim=1)
                embds = torch.cat([tv["embds"], embds], dim=1)
            if tv["task"] == "sot":
                for f_i, ann in enumerate(tv["instances"]):
                    if f_i not in range(first_frame_idx, first_frame_idx + T) or len(ann) == 0:
                        continue
                    ann = ann.to(dev)
                    h, w = ann.image_size
                    upd = [ids.index(i_) for i_ in ann.ori_ids]
                    boxes[upd, f_i] = ann.gt_boxes.tensor / norm
                    rel = -(first_frame_idx + T - f_i)
                    gm = ann.gt_masks.tensor if hasattr(ann.gt_masks, "tensor") else ann.gt_masks
                    masks[upd, rel, :h, :w] = gm.float()
                    mask_logits[upd, rel, :h, :w] = gm.float()
                    labels[upd] = ann.gt_classes          # objects may enter in intermediate frames
                    first_seen[upd] = f_i
            tv.update({"labels": labels, "masks": masks, "mask_logits": mask_logits, "boxes": boxes, "embds": embds,
                       "first_appear_frame_idxs": first_seen})

    # ------------------------------------------------------------------------------------------
    def write_predictions_into_annotations_per_clip(self, out, image_size, targets, first_frame_idx, stride):
        tv = targets[0]
        logits = out["pred_logits"][0].float().sigmoid()   # Q x K (kept for parity of the interface)
        masks = out["pred_masks"][0].float()               # Q x T x h x w
        embds = out["pred_embds"][0].float()               # Q x T x C
        h_p, w_p = masks.shape[-2:]
        boxes = convert_mask_to_box(masks > 0) / torch.as_tensor([w_p, h_p, w_p, h_p], device=masks.device).view(1, 1, -1)
        T = masks.shape[1]
        task = tv["task"]
        if task == "grounding":
            assert self.prompt_as_queries, "only support prompts as queries for referring segmentation task"
        gt_masks, gt_logits, gt_boxes, gt_embds = tv["masks"], tv["mask_logits"], tv["boxes"], tv["embds"]
        _, _, h_gt, w_gt = gt_masks.shape
        masks = _resize(masks, (h_gt, w_gt))
        quality = calculate_mask_quality_scores(masks[..., : image_size[0], : image_size[1]])
        viposeg = "viposeg" in tv["dataset_name"]
        sem_mask = None
        if viposeg and self.use_semseg_pvos:
            n_cls, start = self.dataset_category_info["vipseg"]
            cls_q = logits[..., start:start + n_cls] * quality.view(-1, 1)
            sem_mask = torch.einsum("qc,qthw->cthw", cls_q[: self.num_queries], masks[: self.num_queries].sigmoid()).argmax(0)
        labels = tv["labels"]
        mode = self.video_unified_inference_queries
        with_prompt = self.prompt_as_queries and mode in ("prompt", "prompt+learn", "learn+prompt")
        with_learn = mode in ("learn", "prompt+learn", "learn+prompt")
        first_seen = tv["first_appear_frame_idxs"]

        # ---- (1) objects whose annotation lies inside this clip
        newly = (first_seen >= first_frame_idx) & (first_seen < first_frame_idx + T)
        if newly.any():
            obj = torch.nonzero(newly).reshape(-1)
            faf = first_seen[newly] - (first_frame_idx + T)            # negative frame offsets
            rng = torch.arange(len(obj), device=obj.device)
            prompt_only = task == "sot"
            if prompt_only or with_prompt:
                idx_p = obj + self.num_queries
            gm_first = gt_masks[obj, faf]
            gb_first = gt_boxes[obj, faf]
            if not prompt_only and with_learn:
                # re-identification among the learnable queries: top-5 by box IoU, then the best mask IoU
                biou = video_box_iou(gb_first[:, None].repeat(1, T, 1), boxes)[0][rng, :, faf]
                top = torch.topk(biou, k=5, dim=-1)[1]
                cand = masks[top.flatten(), faf[:, None].repeat(1, 5).flatten()].reshape(-1, 5, h_gt, w_gt).gt(0.0)
                miou = batched_pair_mask_iou(gm_first.unsqueeze(1).repeat(1, 5, 1, 1), cand)
                idx_l = top[rng, miou.argmax(-1)]
            if prompt_only or (self.prompt_as_queries and mode == "prompt"):
                m_masks, m_q, m_embds, m_boxes = masks[idx_p], quality[idx_p], embds[idx_p], boxes[idx_p]
            elif mode == "learn":
                m_masks, m_q, m_embds, m_boxes = masks[idx_l], quality[idx_l], embds[idx_l], boxes[idx_l]
            else:
                den = (quality[idx_p] + quality[idx_l]).clamp(min=1e-5)
                w_pq, w_lq = quality[idx_p] / den, quality[idx_l] / den
                m_masks = w_pq.view(-1, 1, 1, 1) * masks[idx_p] + w_lq.view(-1, 1, 1, 1) * masks[idx_l]
                m_q = calculate_mask_quality_scores(m_masks)
                m_embds = w_pq.view(-1, 1, 1) * embds[idx_p] + w_lq.view(-1, 1, 1) * embds[idx_l]
                m_boxes = w_pq.view(-1, 1, 1) * boxes[idx_p] + w_lq.view(-1, 1, 1) * boxes[idx_l]
            gt_embds[newly, -T:] = m_embds
            if task == "sot":
                is_bg = (m_masks <= 0).all(0)
                weighted = m_masks.sigmoid()
                miou = batched_pair_mask_iou(gm_first.unsqueeze(1), m_masks[rng, faf].gt(0.0).unsqueeze(1)).squeeze(1)
                weighted = weighted * (miou ** 2 * m_q).view(-1, 1, 1, 1)
                owner = weighted.argmax(0)
                owner = torch.where(is_bg, torch.full_like(owner, -1), owner)
                binary = (owner[None] == torch.arange(len(obj), device=owner.device).view(-1, 1, 1, 1)).float()
                m_masks = m_masks * binary
                miou = batched_pair_mask_iou(gm_first.unsqueeze(1), binary[rng, faf].unsqueeze(1)).squeeze(1)
                area = gm_first.flatten(1).sum(1) / (96 * 96)
                ok = miou > 0.15 * area.clamp(max=1)
            else:
                ok = torch.ones(len(obj), dtype=torch.bool, device=obj.device)
            for i_, (ok_i, o_i, f_i) in enumerate(zip(ok.tolist(), obj.tolist(), faf.tolist())):
                f_i = f_i + 1 if task == "sot" else f_i          # the annotated frame itself keeps its annotation
                label = int(labels[o_i])
                is_stuff = viposeg and (label + 1) in self.stuff_dataset_ids
                if (not ok_i and not is_stuff) or f_i == 0:
                    continue
                cur = m_masks[i_, f_i:]
                if is_stuff and sem_mask is not None:            # stuff regions follow the semantic map
                    cur[sem_mask[f_i:] == label] = 10.0
                gt_masks[o_i, f_i:] = cur.gt(0.0)
                gt_logits[o_i, f_i:] = cur
                gt_boxes[o_i, f_i:] = m_boxes[i_, f_i:]

        # ---- (2) objects followed from earlier clips
        seen = (first_seen < first_frame_idx) & (first_seen != -1)
        if seen.any():
            tgt = gt_embds[seen, -self.num_prev_frames_memory:]
            if with_prompt:
                idx_p = torch.nonzero(seen).reshape(-1) + self.num_queries
                cons, sim_p = check_consistency_with_prev_frames(tgt, embds[idx_p], sim_threshold=0.5, return_similarity=True)
                keep = cons.view(-1, 1, 1, 1).float()
                masks_p, q_p = masks[idx_p] * keep, quality[idx_p] * cons.float()
                embds_p, boxes_p = embds[idx_p] * keep.view(-1, 1, 1), boxes[idx_p] * keep.view(-1, 1, 1)
                sim_p = sim_p * cons.float()
            if with_learn:
                use_norm = not viposeg
                idx_l, sim_l = match_from_learnable_embds(tgt, embds[: self.num_queries], return_similarity=True,
                                                          return_src_indices=False, use_norm=use_norm)
                idx_l = torch.as_tensor(idx_l, device=masks.device)
                cons = sim_l >= (0.65 if use_norm else 0.5)
                keep = cons.view(-1, 1, 1, 1).float()
                masks_l, q_l = masks[idx_l] * keep, quality[idx_l] * cons.float()
                embds_l, boxes_l = embds[idx_l] * keep.view(-1, 1, 1), boxes[idx_l] * keep.view(-1, 1, 1)
                sim_l = sim_l * cons.float()
            assert with_prompt or with_learn, "Must use at least one of prompt or learn queries"
            if with_prompt and with_learn:
                sim = (sim_p + sim_l) / (sim_p.gt(0.0).float() + sim_l.gt(0.0).float()).clamp(min=1)
                den = (sim_p + sim_l).clamp(min=1e-5)
                w_pq, w_lq = sim_p / den, sim_l / den
                inter = count_true(masks_p.gt(0) & masks_l.gt(0))
                union = count_true(masks_p.gt(0) | masks_l.gt(0))
                disagree = inter / union.clamp(min=1) < 0.5          # the two sources see different things: trust the prompt
                w_pq = torch.where(disagree, torch.ones_like(w_pq), w_pq)
                w_lq = torch.where(disagree, torch.zeros_like(w_lq), w_lq)
                m_masks = w_pq.view(-1, 1, 1, 1) * masks_p + w_lq.view(-1, 1, 1, 1) * masks_l
                m_q = calculate_mask_quality_scores(m_masks)
                m_embds = w_pq.view(-1, 1, 1) * embds_p + w_lq.view(-1, 1, 1) * embds_l
                m_boxes = w_pq.view(-1, 1, 1) * boxes_p + w_lq.view(-1, 1, 1) * boxes_l
            elif with_prompt:
                sim, m_masks, m_q, m_embds, m_boxes = sim_p, masks_p, q_p, embds_p, boxes_p
            else:
                sim, m_masks, m_q, m_embds, m_boxes = sim_l, masks_l, q_l, embds_l, boxes_l
            if task == "sot":
                # every pixel to the object with the highest sim^2 x quality x probability; drop objects that keep
                # less than a quarter of their own area
                orig = count_true(m_masks > 0).clamp(min=1)
                prob = m_masks.sigmoid()
                if sem_mask is not None:
                    for i_, label in enumerate(labels[seen].tolist()):
                        if (int(label) + 1) in self.stuff_dataset_ids:
                            region = sem_mask == int(label)
                            prob[i_][region] = 1
                            m_masks[i_][region] = 10
                is_bg = (m_masks <= 0).all(0)
                owner = (prob * (sim ** 2 * m_q).view(-1, 1, 1, 1)).argmax(0)
                owner = torch.where(is_bg, torch.full_like(owner, -1), owner)
                binary = (owner[None] == torch.arange(m_masks.shape[0], device=owner.device).view(-1, 1, 1, 1)).float()
                area = count_true(binary)                 # (0 / 1 floats: exact)
                ok = ((area / orig) > 0.25) & (orig > 0) & (area > 0)
                m_masks = m_masks * binary * ok.view(-1, 1, 1, 1).float()
            gt_logits[seen, -T:] += m_masks
            gt_boxes[seen, -T:] = m_boxes
            cur = gt_embds[seen, -T:]
            gt_embds[seen, -T:] = (cur + m_embds) / ((cur != 0).any(-1)[..., None] + 1.0)

        tv["masks"] = gt_logits.gt(0.0).float()
        tv["mask_logits"] = gt_logits
        tv["boxes"] = gt_boxes
        tv["embds"] = gt_embds

    # ------------------------------------------------------------------------------------------
    def _finished_frames(self, tv, first_frame_idx, image_size, out_size, is_last, stride):
        video_len = tv["video_len"]
        T = min(self.num_frames, video_len - first_frame_idx)
        m = tv["mask_logits"]
        m = m[:, -T:] if is_last else m[:, -T:min(-T + stride, -1)]
        m = m[:, :, : image_size[0], : image_size[1]]
        if tuple(image_size) != tuple(out_size):
            m = _resize(m.float(), out_size)
        return m.gt(0.0).float()

    def save_vos_results(self, first_frame_idx, targets, image_size, out_size, is_last, stride):
        """Object-id map of the frames this clip finishes: [T', H_out, W_out] uint8 on the host (0 = background)."""
        tv = targets[0]
        ids = torch.as_tensor(tv["ids"], device=self.device)
        if ids.numel() and ids.min() == 0:
            ids = ids + 1                      # RefDAVIS numbers its expressions from 0
        m = self._finished_frames(tv, first_frame_idx, image_size, out_size, is_last, stride).transpose(0, 1)   # T' N H W
        if m.shape[1] == 0:
            return torch.zeros(m.shape[0], *m.shape[-2:], dtype=torch.uint8)
        idmap = ids[m.argmax(1)]
        idmap = torch.where((m <= 0).all(1), torch.zeros_like(idmap), idmap)
        return idmap.to(torch.uint8).cpu()

    def save_rvos_results(self, first_frame_idx, targets, image_size, out_size, is_last, stride):
        """Per-expression binary masks of the frames this clip finishes: {"ids": [...], "masks": [N, T', H, W] uint8}."""
        tv = targets[0]
        m = self._finished_frames(tv, first_frame_idx, image_size, out_size, is_last, stride)
        return {"ids": list(tv["ids"]), "masks": (m * 255).to(torch.uint8).cpu()}
