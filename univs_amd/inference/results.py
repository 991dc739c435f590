"""Result formats of the clip loops: what the reference's evaluators read from disk.

    rle_encode_masks / rle_decode / rle_area      COCO run-length masks.  The reference calls `pycocotools.mask.encode`
                                                  once per object and frame on host copies of full-resolution masks
                                                  (inference_video_entity.py:944-948); here the run boundaries of ALL masks
                                                  are found on the device in one pass (column-major difference + one
                                                  `nonzero`), only the boundary positions cross PCIe, and the
                                                  variable-length string coding runs vectorised over all runs on the host.
    calculate_mask_temporal_consistency_scores    univs/inference/comm.py:197-207
    vis_clip_instances_to_coco_json_video         univs/inference/comm.py:97-195   per-video YouTube-VIS style records
    write_vos_pngs / write_rvos_pngs              inference_video_vos.py:622-705   palette id maps / per-expression masks
    write_vps_predictions / write_vps_json        univs/evaluation/vps_evaluation.py:117-205   VIPSeg layout: `pan_pred/<video>/<frame>.png`
                                                  colour-coded panoptic maps + the `pred.json` segment records that VPQ / STQ read
    write_vss_predictions                         univs/evaluation/vss_evaluation.py:93-118    VSPW layout: `<video>/<frame>.png` class ids

pycocotools is a third-party dependency of the reference (un-pinned: INSTALL.md installs the latest release) and is absent
from this image, so the RLE string coding is restated from its published algorithm (`rleToString` / `rleFrString` in
maskApi.c: 5 payload bits + continuation bit per character, offset 48, counts after the third stored as differences to the
count two places back) -- parity with the library is UNPINNED; what is tested: encode/decode round trips, hand-derived
known-answer strings, and the equality of the device and host boundary finders.
"""
import os
from typing import List, Sequence

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------------------------
# COCO RLE
# ------------------------------------------------------------------------------------------------------------------
def _counts_to_string(counts: np.ndarray, starts: np.ndarray) -> List[str]:
    """counts: int64 run lengths of all masks back to back; starts: [n_masks + 1] offsets into it.  -> one string per mask."""
    n = counts.shape[0]
    if n == 0:
        return ["" for _ in range(len(starts) - 1)]
    pos = np.arange(n) - np.repeat(starts[:-1], np.diff(starts))          # index of the run inside its mask
    x = counts.astype(np.int64).copy()
    delta = pos > 2
    x[delta] -= counts[np.nonzero(delta)[0] - 2]
    # up to 13 characters of 5 bits for a 64-bit value; run lengths of images need at most 7
    chars = np.zeros((n, 13), dtype=np.uint8)
    length = np.zeros(n, dtype=np.int64)
    alive = np.ones(n, dtype=bool)
    k = 0
    while alive.any():
        c = x & 0x1F
        x = x >> 5                                                       # arithmetic shift: negative deltas end at -1
        more = np.where((c & 0x10) != 0, x != -1, x != 0) & alive
        c = np.where(more, c | 0x20, c) + 48
        chars[alive, k] = c[alive].astype(np.uint8)
        length[alive] = k + 1
        alive = more
        k += 1
    keep = np.arange(13)[None, :] < length[:, None]
    flat = chars[keep].tobytes().decode("ascii")
    per_run_end = np.cumsum(length)
    ends = np.concatenate([[0], per_run_end])[starts]
    return [flat[ends[i]:ends[i + 1]] for i in range(len(starts) - 1)]


def mask_run_lengths(masks: torch.Tensor):
    """masks: [N, H, W] (bool / 0-1) on any device -> (counts int64 numpy, starts [N + 1]) of the column-major runs,
    each mask's list beginning with a run of zeros (possibly of length 0), as COCO defines it."""
    N, H, W = masks.shape
    hw = H * W
    if N == 0 or hw == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(N + 1, dtype=np.int64)
    m = masks.bool().transpose(1, 2).reshape(N, hw)
    change = m[:, 1:] != m[:, :-1]
    # a leading foreground pixel opens with an empty run of zeros: treat it as a change at position 0
    lead = m[:, :1]
    idx = torch.nonzero(torch.cat([lead, change], dim=1))                # [R, 2] sorted by mask, then position (one sync)
    idx = idx.cpu().numpy()
    rows, posn = idx[:, 0], idx[:, 1]
    per_mask = np.bincount(rows, minlength=N)
    # boundaries of mask i: its change positions, then hw; run k = boundary k - boundary k-1 (first from 0)
    n_runs = per_mask + 1
    starts = np.concatenate([[0], np.cumsum(n_runs)]).astype(np.int64)
    bounds = np.empty(int(starts[-1]), dtype=np.int64)
    last = starts[1:] - 1
    bounds[last] = hw
    body = np.ones(bounds.shape[0], dtype=bool)
    body[last] = False
    bounds[body] = posn
    prev = np.empty_like(bounds)
    prev[1:] = bounds[:-1]
    prev[starts[:-1]] = 0
    return bounds - prev, starts


def rle_encode_masks(masks: torch.Tensor) -> List[dict]:
    """[N, H, W] binary masks -> N dicts {"size": [H, W], "counts": str} (compressed COCO RLE, counts already `str`)."""
    if masks.dim() == 2:
        masks = masks[None]
    N, H, W = masks.shape
    counts, starts = mask_run_lengths(masks)
    return [{"size": [int(H), int(W)], "counts": s} for s in _counts_to_string(counts, starts)]


def rle_counts(rle: dict) -> np.ndarray:
    """The run lengths stored in a compressed RLE string."""
    s = rle["counts"]
    s = s.decode("ascii") if isinstance(s, (bytes, bytearray)) else s
    out, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(out) > 2:
            x += out[-2]
        out.append(x)
    return np.asarray(out, dtype=np.int64)


def rle_decode(rle: dict) -> np.ndarray:
    """-> uint8 [H, W]."""
    H, W = rle["size"]
    counts = rle_counts(rle)
    vals = (np.arange(len(counts)) % 2).astype(np.uint8)
    flat = np.repeat(vals, counts)
    if flat.shape[0] != H * W:
        raise ValueError(f"RLE covers {flat.shape[0]} pixels, the mask has {H * W}")
    return flat.reshape(W, H).T.copy()


def rle_area(rle: dict) -> int:
    return int(rle_counts(rle)[1::2].sum())


# ------------------------------------------------------------------------------------------------------------------
# YouTube-VIS style per-video records
# ------------------------------------------------------------------------------------------------------------------
def calculate_mask_temporal_consistency_scores(scores: torch.Tensor) -> torch.Tensor:
    """scores [n_clips, K] of one entity, in place: a clip's scores count only if the entity is non-blank there and --
    with the reference's window arithmetic (dt = 1, half-open slice) -- also in the clip before it."""
    nonblank = scores.sum(-1) > 0
    n = len(nonblank)
    for t in range(n):
        s_t, e_t = max(0, t - 1), min(n, t + 1)
        scores[t] *= nonblank[t] * nonblank[s_t:e_t].sum() / max(e_t - s_t, 1)
    return scores


def vis_clip_instances_to_coco_json_video(batched_inputs, results_list, apply_cls_thresh=0.05, test_topk_per_video=25):
    """Merge the per-clip records of `InferenceVideoEntity.save_results_vis` into one video's result list.

    results_list: list (clips) of lists of {"obj_id", "score" [K], "frame_id_start", "masks" bool [T', H, W] (ours) or
    "segmentations" (list of RLE dicts, the reference's), optional "mask_quality_score"}.
    -> list of {"video_id", "score", "category_id", "segmentations" (one RLE per frame), "height", "width"}; one record per
    (entity, class) whose score passes a tenth of the class threshold, cut to the top max(1.5 x #confident, top-k)."""
    assert len(batched_inputs) == 1, "More than one inputs are loaded for inference!"
    info = batched_inputs[0]
    try:
        video_id = int(info["video_id"])
    except (TypeError, ValueError):
        video_id = info["video_id"]
    video_len, height, width = int(info["video_len"]), int(info["height"]), int(info["width"])
    blank = rle_encode_masks(torch.zeros(1, height, width, dtype=torch.bool))[0]

    # encode every clip's masks in one device pass per record list
    for clip in results_list:
        todo = [r for r in clip if "segmentations" not in r]
        if todo:
            stacked = torch.cat([r["masks"] for r in todo], dim=0)
            rles = rle_encode_masks(stacked)
            o = 0
            for r in todo:
                n = r["masks"].shape[0]
                r["segmentations"] = rles[o:o + n]
                o += n

    out, out_scores, confident = [], [], 0
    obj_ids = set(r["obj_id"] for clip in results_list for r in clip)
    for obj_id in obj_ids:
        segm = [blank] * video_len
        cls_scores, quality = [], []
        for clip in results_list:
            for r in clip:
                if r["obj_id"] != obj_id:
                    continue
                if "mask_quality_score" in r:
                    quality.append(r["mask_quality_score"])
                cls_scores.append(torch.as_tensor(r["score"]).float().cpu())
                s = r["frame_id_start"]
                segm[s:s + len(r["segmentations"])] = r["segmentations"]
        assert len(segm) == video_len, f"The video has {video_len} frames, but the prediction has {len(segm)} frames!"
        assert len(cls_scores), "Miss category scores here!"
        scores = torch.stack(cls_scores, dim=0)
        if quality:
            q = sum(quality) / len(quality)
        else:
            q = ((scores.sum(-1) > 0).sum(0) / video_len).clamp(min=0.1)
        scores = calculate_mask_temporal_consistency_scores(scores)
        scores = scores.sum(0) / (scores.sum(-1) > 0).sum(0).clamp(min=1)
        for c, sc in enumerate(scores.tolist()):
            if sc < 0.1 * apply_cls_thresh:
                continue
            s = sc * float(q)
            out.append({"video_id": video_id, "score": s, "category_id": c, "segmentations": segm, "height": height,
                        "width": width})
            out_scores.append(s)
            confident += sc > apply_cls_thresh
    if out_scores:
        ranked = sorted(out_scores, reverse=True)
        cut = ranked[min(max(int(confident * 1.5), test_topk_per_video), len(ranked) - 1)]
        out = [r for r in out if r["score"] >= cut]
    return out


# ------------------------------------------------------------------------------------------------------------------
# VOS / RefVOS png files
# ------------------------------------------------------------------------------------------------------------------
def _png_name(file_name: str) -> str:
    return file_name.split("/")[-1].replace(".jpg", ".png")


def write_vos_pngs(output_dir: str, file_names: Sequence[str], first_frame_idx: int, idmaps: torch.Tensor, palette=None):
    """`InferenceVideoVOS.save_vos_results` output ([T', H, W] uint8 object ids) -> palette PNGs
    `<output_dir>/inference/Annotations/<video>/<frame>.png`, the layout the DAVIS / YouTube-VOS servers expect."""
    from PIL import Image
    save_dir = os.path.join(output_dir, "inference/Annotations", file_names[0].split("/")[-2])
    os.makedirs(save_dir, exist_ok=True)
    paths = []
    for t, m in enumerate(idmaps.cpu().numpy().astype(np.uint8)):
        img = Image.fromarray(m)
        if palette is not None:
            img.putpalette(palette)
        paths.append(os.path.join(save_dir, _png_name(file_names[first_frame_idx + t])))
        img.save(paths[-1])
        img.close()
    return paths


def write_rvos_pngs(output_dir: str, file_names: Sequence[str], first_frame_idx: int, result: dict):
    """`InferenceVideoVOS.save_rvos_results` output ({"ids", "masks" [N, T', H, W] uint8 0/255}) -> grey PNGs
    `<output_dir>/inference/Annotations/<video>/<expression id>/<frame>.png` (Ref-YouTube-VOS / Ref-DAVIS layout)."""
    from PIL import Image
    video = file_names[0].split("/")[-2]
    paths = []
    for id_, mi in zip(result["ids"], result["masks"].cpu().numpy().astype(np.uint8)):
        save_dir = os.path.join(output_dir, "inference/Annotations", video, str(id_))
        os.makedirs(save_dir, exist_ok=True)
        for t, m in enumerate(mi):
            img = Image.fromarray(m)
            paths.append(os.path.join(save_dir, _png_name(file_names[first_frame_idx + t])))
            img.save(paths[-1])
            img.close()
    return paths


# ------------------------------------------------------------------------------------------------------------------
# VPS (VIPSeg) and VSS (VSPW) result files
# ------------------------------------------------------------------------------------------------------------------
def rgb2id(color):
    """panopticapi.utils.rgb2id (COCO panoptic format: id = R + 256 G + 256^2 B), for one colour or an [..., 3] array."""
    if isinstance(color, np.ndarray) and color.ndim >= 1 and color.shape[-1] == 3 and color.ndim > 1:
        c = color.astype(np.int32)
        return c[..., 0] + 256 * c[..., 1] + 256 * 256 * c[..., 2]
    return int(color[0]) + 256 * int(color[1]) + 256 * 256 * int(color[2])


class IdGenerator:
    """panopticapi.utils.IdGenerator, restated from its published algorithm (cocodataset/panopticapi, utils.py; the package is a
    third-party dependency of the reference's VPSEvaluator and absent from this image: parity with the library UNPINNED): stuff
    categories always get their own colour; the first segment of a thing category gets the category colour, further segments a
    random colour within +-30 per channel of it (numpy's global generator, as the library) that is not taken yet."""

    def __init__(self, categories):
        self.taken_colors = set([0, 0, 0])
        self.categories = categories
        for category in self.categories.values():
            if category["isthing"] == 0:
                self.taken_colors.add(tuple(category["color"]))

    def get_color(self, cat_id):
        def random_color(base, max_dist=30):
            new_color = base + np.random.randint(low=-max_dist, high=max_dist + 1, size=3)
            return tuple(np.maximum(0, np.minimum(255, new_color)))
        category = self.categories[cat_id]
        if category["isthing"] == 0:
            return category["color"]
        base_color_array = category["color"]
        base_color = tuple(base_color_array)
        if base_color not in self.taken_colors:
            self.taken_colors.add(base_color)
            return base_color
        while True:
            color = random_color(base_color_array)
            if color not in self.taken_colors:
                self.taken_colors.add(color)
                return color


def write_vps_predictions(inputs: dict, outputs: dict, output_dir: str, categories: dict):
    """One video's `vps_output_results` dict ({"image_size", "pred_masks" [T, H, W] segment ids, "segments_infos": [{"id", "isthing",
    "category_id"}]}) -> `<output_dir>/pan_pred/<video>/<frame>.png` (RGB panoptic maps) and the video's record for `pred.json`:
    {"annotations": [{"segments_info": [{"bbox", "area", "category_id", "iscrowd", "id"}], "file_name"}], "video_id"} -- what
    `VPSEvaluator.process` writes / collects (univs/evaluation/vps_evaluation.py:117-178; the bbox is [x, y, x_max - x, y_max - y] as
    there).  `inputs`: the batched input of the video ("file_names", "frame_indices"); `categories`: {category id: {"id", "isthing",
    "color"}} (the dataset metadata's `categories`)."""
    from PIL import Image
    color_generator = IdGenerator(categories)
    image_names = [inputs["file_names"][int(i)] for i in inputs["frame_indices"]]
    video_id = image_names[0].split("/")[-2]
    H, W = int(outputs["image_size"][0]), int(outputs["image_size"][1])
    pan = outputs["pred_masks"]
    pan = pan.cpu().numpy() if isinstance(pan, torch.Tensor) else np.asarray(pan)
    pan_format = np.zeros((pan.shape[0], H, W, 3), dtype=np.uint8)
    per_segment = []
    for info in outputs["segments_infos"]:
        sem = info["category_id"]
        mask = pan == info["id"]
        color = color_generator.get_color(sem)
        pan_format[mask] = color
        base = {"category_id": int(sem) - 1, "iscrowd": 0, "id": int(rgb2id(color))}
        dts = []
        for i in range(pan.shape[0]):
            ys, xs = np.where(mask[i])
            if len(ys) == 0:
                dts.append(None)
                continue
            x, y = xs.min(), ys.min()
            dts.append(dict({"bbox": [int(x), int(y), int(xs.max() - x), int(ys.max() - y)], "area": int(mask[i].sum())}, **base))
        per_segment.append(dts)
    save_dir = os.path.join(output_dir, "pan_pred", video_id)
    os.makedirs(save_dir, exist_ok=True)
    annotations = []
    for i, name in enumerate(image_names):
        img = Image.fromarray(pan_format[i])
        img.save(os.path.join(save_dir, name.split("/")[-1].split(".")[0] + ".png"))
        img.close()
        annotations.append({"segments_info": [d[i] for d in per_segment if d[i] is not None], "file_name": name.split("/")[-1]})
    return {"annotations": annotations, "video_id": video_id}


def write_vps_json(predictions: Sequence[dict], output_dir: str) -> str:
    """`VPSEvaluator.evaluate`'s file (vps_evaluation.py:196-199): {"annotations": [one record per video]} -> `<output_dir>/pred.json`."""
    import json
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, "pred.json")
    with open(path, "w") as f:
        json.dump({"annotations": list(predictions)}, f)
    return path


def write_vss_predictions(inputs: dict, outputs: dict, output_dir: str, contiguous_id_to_dataset_id: dict, ignore_val: int = 255):
    """One video's `vss_output_results` dict ({"image_size", "pred_masks" [T, H, W] contiguous class ids}) -> `<output_dir>/<video_id>/
    <frame>.png`, uint8 class ids as VSPW's evaluation reads them: the dataset id of the class minus the smallest dataset id, 255 where
    the prediction is the ignore value (`VSSEvaluator.process`, univs/evaluation/vss_evaluation.py:93-118)."""
    from PIL import Image
    video_id = str(inputs["video_id"])
    image_names = [inputs["file_names"][int(i)] for i in inputs["frame_indices"]]
    sem = outputs["pred_masks"]
    sem = (sem.cpu().numpy() if isinstance(sem, torch.Tensor) else np.asarray(sem)).astype(np.uint8)
    out = np.full_like(sem, 255, dtype=np.uint8)
    lo = min(contiguous_id_to_dataset_id.values())
    for cls_id in np.unique(sem):
        if cls_id == ignore_val:
            continue
        out[sem == cls_id] = contiguous_id_to_dataset_id[int(cls_id)] - lo
    assert len(image_names) == len(out), "Mismatch length between predicted and gt images"
    save_dir = os.path.join(output_dir, video_id)
    os.makedirs(save_dir, exist_ok=True)
    paths = []
    for i, name in enumerate(image_names):
        img = Image.fromarray(out[i])
        paths.append(os.path.join(save_dir, name.split("/")[-1].split(".")[0] + ".png"))
        img.save(paths[-1])
        img.close()
    return paths
