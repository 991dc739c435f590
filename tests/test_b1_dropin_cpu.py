"""Boundary B1: the REFERENCE's own callers, imported unchanged from /root/reference through
oracle/ref_harness.py, driving THIS repository's modules (`backbone`, `sem_seg_head`) instead of the reference's -- the
"drops in unchanged above the module boundary" claim of the north star, executed.

  * `InferenceVideoEntity.inference_video` (univs/inference/inference_video_entity.py:301-404: `model.backbone(...)` :312,
    `model.sem_seg_head(features, targets=targets)` :316, and everything it then does with the outputs and the mutated
    `targets` dict) over our Swin + MaskFormerHead -> the per-clip states must equal g11a, which is the same loop over the
    reference's modules;
  * the long-video split call (univs/univs_prompt_longvideo.py:397-406: `pixel_decoder.forward_features(features)` ->
    `predictor(multi_scale_features, mask_features, mask_features_bfe_conv, targets=...)`) and the attributes it reads
    (:562-571: `predictor.pe_layer`, `.input_proj[i]`, `.level_embed.weight`, `.forward_prompt_encoder`);
  * `MaskFormer_Video`'s call (mask2former_video/video_maskformer_model.py:208-209: `sem_seg_head(features)`).

Dev container only (the reference does not travel): skipped where /root/reference is absent.  The four HIP operators
are replaced by the oracle's CPU stand-ins (oracle/cpu_path.py), as in every CPU module test."""
import os
import types

import numpy as np
import pytest
import torch

from oracle.cpu_path import cpu_ops
from tests import cases, helpers

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/univs"), reason="needs the reference checkout (dev container)")


def test_reference_clip_loop_drives_our_modules(golden_dir):
    from oracle import gen_golden                      # imports the reference through oracle/ref_harness.py
    from tests.test_clip_loop_cpu import compare_states
    g = np.load(os.path.join(golden_dir, "g11a_clip_loop_model.npz"))
    case = cases.LOOP_CASE
    ours = types.SimpleNamespace(backbone=helpers.build_swin("cpu"), sem_seg_head=helpers.build_head(case, "cpu"))
    with cpu_ops(), torch.no_grad():
        d = gen_golden._ref_loop(case, ours, stability_score_thresh=0.0)      # the REFERENCE's loop, our modules
    got = {k: torch.as_tensor(v) for k, v in d.items() if not k.startswith("draws_")}
    assert got["clip_first_frames"].tolist() == g["clip_first_frames"].tolist()
    compare_states(got, g, tol=1e-3, mask_margin=1e-3)
    # the generator's draw recorder hooks OUR sampler here (same method names as the reference's encoder): with the same
    # seed our sampler picks the very pixels the reference picked
    for k in ("draws_n", "draws_clip", "draws_point_idx", "draws_feat_idx"):
        assert np.array_equal(np.asarray(d[k]), g[k]), k


def test_long_video_split_call_and_attributes():
    """univs_prompt_longvideo.py:397-406, :562-571 against the reference's modules on the same inputs."""
    from oracle import gen_golden, ref_harness as rh
    R = rh.ref()
    case = cases.HEAD_CASE
    feats = cases.backbone_features(case)
    ours = helpers.build_head(case, "cpu", return_aux=False)
    theirs = gen_golden._ref_head(R, case)
    with cpu_ops(), torch.no_grad():
        t_o = ours.pixel_decoder.forward_features(feats)
        t_r = theirs.pixel_decoder.forward_features(feats)
        assert len(t_o) == len(t_r) == 4
        for a, b in zip(t_o[:3], t_r[:3]):
            assert (a - b).abs().max().item() < 1e-4
        for a, b in zip(t_o[3], t_r[3]):
            assert (a - b).abs().max().item() < 1e-4
        mask_features, mask_features_bfe_conv, multi_scale = t_o[0], t_o[1], t_o[-1]
        tg_o, tg_r = cases.targets_first_clip(case), cases.targets_first_clip(case)
        out_o = ours.predictor(multi_scale, mask_features, mask_features_bfe_conv, targets=tg_o)
        out_r = theirs.predictor(t_r[-1], t_r[0], t_r[1], targets=tg_r)
        whole = ours(feats, targets=cases.targets_first_clip(case))
    for k in ("pred_logits", "pred_masks", "pred_embds"):
        assert (out_o[k] - out_r[k]).abs().max().item() < 1e-3, k
        assert torch.equal(out_o[k], whole[k]), k           # the split call IS the head's forward
    # the attributes the long-video model reads directly
    p_o, p_r = ours.predictor, theirs.predictor
    x = multi_scale[0]
    bs, t = 1, case["T"]
    pos_o = p_o.pe_layer(x.view(bs, t, -1, *x.shape[-2:]), None)
    pos_r = p_r.pe_layer(x.view(bs, t, -1, *x.shape[-2:]), None)
    assert (pos_o - pos_r).abs().max().item() < 1e-5
    for i in range(3):
        assert torch.equal(p_o.input_proj[i](multi_scale[i]), p_r.input_proj[i](multi_scale[i]))
    assert torch.equal(p_o.level_embed.weight, p_r.level_embed.weight)
    assert callable(p_o.forward_prompt_encoder) and ours.num_classes == theirs.num_classes


def test_maskformer_video_is_registered_and_calls_the_head_without_targets():
    """video_maskformer_model.py:208-209: `features = self.backbone(images.tensor); outputs = self.sem_seg_head(features)`."""
    import univs_amd.modeling.meta_arch.univs_prompt  # noqa: F401  (registers the META_ARCH classes, as build_model does)
    from univs_amd.registry import META_ARCH_REGISTRY
    assert "MaskFormer_Video" in META_ARCH_REGISTRY
    from univs_amd.modeling.meta_arch.univs_prompt import MaskFormer_Video
    case = cases.HEAD_CASE
    head = helpers.build_head(case, "cpu", return_aux=False)
    with cpu_ops(), torch.no_grad(), pytest.raises(ValueError):
        head(cases.backbone_features(case))          # no targets and no opt-in: raises, as the reference's decoder does
    calls = []

    class Backbone(torch.nn.Module):
        size_divisibility = 32

        def forward(self, x):
            calls.append(tuple(x.shape))
            return cases.backbone_features(case)
    m = MaskFormer_Video(backbone=Backbone(), sem_seg_head=head, num_frames=case["T"], size_divisibility=32,
                         pixel_mean=(123.675, 116.28, 103.53), pixel_std=(58.395, 57.12, 57.375)).eval()
    frames = [torch.zeros(3, case["H"], case["W"]) for _ in range(case["T"])]
    with cpu_ops(), torch.no_grad():
        out = m([{"image": frames, "height": case["H"], "width": case["W"]}])
    assert calls == [(case["T"], 3, case["H"], case["W"])]
    assert set(out) >= {"pred_logits", "pred_masks"} and out["pred_masks"].shape[2] == case["T"]
