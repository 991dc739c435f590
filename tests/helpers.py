"""Builders of the PRODUCT modules with the closed-form weights, shared by the CPU and GPU module tests."""
import numpy as np
import torch

from tests import cases
from univs_amd import synth
from univs_amd.registry import ShapeSpec


def build_swin(device="cpu", variant=None):
    from univs_amd.modeling.backbone.swin import SwinTransformer
    k = dict(variant or cases.SWIN_T)
    m = SwinTransformer(k["pretrain_img_size"], k["patch_size"], k["in_chans"], k["embed_dim"], k["depths"],
                        k["num_heads"], k["window_size"], k["mlp_ratio"], k["qkv_bias"], k["qk_scale"], k["ape"],
                        k["patch_norm"]).eval()
    synth.load_synthetic(m, prefix="backbone.")
    return m.to(device)


def build_pixel_decoder(shapes, device="cpu"):
    from univs_amd.modeling.pixel_decoder.msdeformattn import MSDeformAttnPixelDecoder
    ish = {k: ShapeSpec(channels=c, stride=s) for k, (c, s) in shapes.items()}
    m = MSDeformAttnPixelDecoder(ish, **cases.PIXDEC).eval()
    synth.load_synthetic(m, prefix="sem_seg_head.pixel_decoder.")
    return m.to(device)


def build_head(case, device="cpu", return_aux=True, **dec_over):
    from univs_amd.modeling.meta_arch.mask_former_head import MaskFormerHead
    from univs_amd.modeling.prompt_encoder import VisualPromptSampler
    from univs_amd.modeling.transformer_decoder.univs_decoder import VideoMultiScaleMaskedTransformerDecoderUniVS
    pd = build_pixel_decoder(case["shapes"])
    ish = {k: ShapeSpec(channels=c, stride=s) for k, (c, s) in case["shapes"].items()}
    dec = VideoMultiScaleMaskedTransformerDecoderUniVS(
        clip_class_embed_path=cases.clip_table(), visual_prompt_sampler=VisualPromptSampler(**cases.sampler_kwargs(case)),
        return_aux_outputs=return_aux, **cases.decoder_kwargs(case, **dec_over)).eval()
    synth.load_synthetic(dec, prefix="sem_seg_head.predictor.")
    head = MaskFormerHead(ish, num_classes=133, pixel_decoder=pd, pixel_decoder_name="MSDeformAttnPixelDecoder",
                          transformer_predictor=dec, transformer_in_feature="multi_scale_pixel_decoder").eval()
    return head.to(device)


HEAD_SCENARIOS = [
    ("g6_head_first_clip", {}, cases.targets_first_clip, None),
    ("g7_head_visual_prompts", {}, cases.targets_with_entities, 0),
    ("g8_head_grounding", dict(text_to_image=True, sa_mask="sep-blocked"), cases.targets_grounding, None),
    ("g8b_head_detection_text", dict(text_to_image=True),
     lambda: cases.targets_first_clip(prompt_type="text", dataset="vspw"), None),
]


def advance_to_third_clip(targets):
    """Same manipulation of the caller-owned dict as oracle/gen_golden.py:g7_head_visual_prompts."""
    tv = targets[0]
    T = cases.HEAD_CASE["T"]
    tv["first_frame_idx"] = 2
    tv["frame_indices"] = torch.arange(2, 2 + T)
    tv["masks"] = torch.cat([tv["masks"], torch.zeros_like(tv["masks"][:, :1])], 1)
    tv["masks"][:, -2] = tv["masks"][:, -3]
    tv["boxes"] = torch.cat([tv["boxes"], torch.zeros_like(tv["boxes"][:, :1])], 1)
    tv["boxes"][:, -2] = tv["boxes"][:, -3]
    tv["ids"] = torch.cat([tv["ids"], tv["ids"][:, :1]], 1)


def check_head_outputs(out, g, prefix, tol):
    """pred_masks within `tol` max-abs and sign-identical away from 0 (the north-star contract: 1e-3,
    argmax/>0 identical); class logits and embeddings likewise."""
    stats = {}
    for k in ("pred_logits", "pred_masks", "pred_embds", "pred_reid_logits"):
        gk = prefix + k
        if gk not in g.files:
            continue
        got = out[k].detach().cpu().numpy()
        ref = g[gk]
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        err = np.abs(got - ref).max()
        stats[k] = err
        # mask logits: the north star's ABSOLUTE bound; class logits / embeddings (not part of it): relative to magnitude
        assert err < (tol if k == "pred_masks" else tol * max(1.0, np.abs(ref).max() / 10.0)), (k, err)
    pm, ref = out["pred_masks"].detach().cpu().numpy(), g[prefix + "pred_masks"]
    flips = ((pm > 0) != (ref > 0)) & (np.abs(ref) > tol)
    assert flips.sum() == 0, f"{flips.sum()} mask sign flips"
    return stats
