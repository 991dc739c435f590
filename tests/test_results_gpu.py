"""GPU: result formats (univs_amd/inference/results.py) fed with DEVICE tensors -- as the clip loops hand them over -- against the
oracle and the goldens the reference's own evaluators wrote: COCO RLE strings == oracle/ops_ref.c (maskApi.c restated) at full
size, the per-video VIS records == g17 (reference merge logic, inference/comm.py:97-195), the VPS / VSS files == g21
(VPSEvaluator / VSSEvaluator of the imported reference).  The CPU counterparts: tests/test_results_cpu.py."""
import json
import os

import numpy as np
import pytest
import torch

from tests import cases
from univs_amd import synth
from univs_amd.inference import results as R

pytestmark = pytest.mark.gpu


def _to(x, dev):
    if isinstance(x, torch.Tensor):
        return x.to(dev)
    if isinstance(x, dict):
        return {k: _to(v, dev) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_to(v, dev) for v in x)
    return x


def test_rle_of_device_masks_matches_the_c_oracle(cuda):
    """results.rle_encode_masks on device masks (run boundaries found on the device) == oracle.c_ops.rle_encode, string for string, at
    736 x 1280: noise (hundreds of thousands of runs), blobs, stripes, empty, full; and at 1088 x 1920 (long codes)."""
    from oracle import c_ops
    m = torch.zeros(6, 736, 1280, dtype=torch.bool)
    m[0] = synth.uniform("rle/noise", (736, 1280)) > 0.3
    m[1, 100:500, 200:900] = True
    m[1, 300:320, 0:1280] = False
    m[2] = (synth.uniform("rle/blobs", (46, 80)) > 0.6).repeat_interleave(16, 0).repeat_interleave(16, 1)
    m[4] = True
    m[5] = (torch.arange(1280) % 3 == 0)[None, :].expand(736, 1280)
    dev = R.rle_encode_masks(m.to(cuda))
    for i, r in enumerate(dev):
        counts, s = c_ops.rle_encode(m[i].numpy())
        assert r["counts"] == s and r["size"] == [736, 1280], i
        assert np.array_equal(R.rle_counts(r), counts)
        assert R.rle_area(r) == int(m[i].sum())
    big = torch.zeros(2, 1088, 1920, dtype=torch.bool)
    big[0, 100:900, 300:1500] = True
    big[1, :, 1919] = True
    for i, r in enumerate(R.rle_encode_masks(big.to(cuda))):
        assert r["counts"] == c_ops.rle_encode(big[i].numpy())[1]


@pytest.mark.parametrize("tag,kw", [("default", {}), ("tight", dict(apply_cls_thresh=0.5, test_topk_per_video=2))])
def test_vis_records_from_device_tensors_match_reference(cuda, golden_dir, tag, kw):
    g = np.load(os.path.join(golden_dir, "g17_vis_results.npz"))
    info, clips = cases.vis_result_records()
    res = R.vis_clip_instances_to_coco_json_video(info, _to(clips, cuda), **kw)
    host = R.vis_clip_instances_to_coco_json_video(info, clips, **kw)
    assert len(res) == len(g[f"{tag}_score"])
    assert [r["category_id"] for r in res] == g[f"{tag}_category"].tolist()
    assert np.allclose([r["score"] for r in res], g[f"{tag}_score"], rtol=1e-6, atol=1e-7)
    areas = np.array([[R.rle_area(s) for s in r["segmentations"]] for r in res])
    assert np.array_equal(areas, g[f"{tag}_areas"])
    assert [[s["counts"] for s in r["segmentations"]] for r in res] == [[s["counts"] for s in r["segmentations"]] for r in host]
    json.dumps(res)                                                                       # json-serialisable: no tensors left


def test_vps_and_vss_files_from_device_tensors_match_reference(cuda, golden_dir, tmp_path):
    from PIL import Image
    g = np.load(os.path.join(golden_dir, "g21_result_files.npz"))
    inputs = cases.result_file_inputs()
    np.random.seed(7)
    rec = R.write_vps_predictions(inputs, _to(cases.vps_result_outputs(), cuda), str(tmp_path / "vps"), cases.VPS_CATEGORIES)
    names = sorted(os.listdir(tmp_path / "vps" / "pan_pred" / "vid_0007"))
    assert names == json.loads(str(g["vps_png_names"]))
    png = np.stack([np.asarray(Image.open(tmp_path / "vps" / "pan_pred" / "vid_0007" / n)) for n in names])
    assert np.array_equal(png, g["vps_png"])
    assert json.dumps(rec, sort_keys=True, default=int) == str(g["vps_record"])
    assert open(R.write_vps_json([rec], str(tmp_path / "vps"))).read() == str(g["vps_pred_json"])
    paths = R.write_vss_predictions(inputs, _to(cases.vss_result_outputs(), cuda), str(tmp_path / "vss"), cases.VSS_CONTIGUOUS_TO_DATASET)
    assert [os.path.basename(p) for p in paths] == json.loads(str(g["vss_png_names"]))
    assert np.array_equal(np.stack([np.asarray(Image.open(p)) for p in paths]), g["vss_png"])
