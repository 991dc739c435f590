"""Builders of the PRODUCT modules with the closed-form weights, shared by the CPU and GPU module tests."""
import numpy as np
import torch

from tests import cases
from univs_amd import synth
from univs_amd.registry import ShapeSpec  # noqa: F401
from univs_amd.workloads import build_head, build_pixel_decoder, build_swin  # noqa: F401


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


def check_g4_g5(head, g, device, ops):
    """SURVEY.md Appendix B G4 / G5 against the fixtures captured inside the reference decoder
    (oracle/gen_golden.py:g4_g5_teacher_forced).  `head` is our MaskFormerHead on `device`; `ops` the operator namespace
    the decoder calls (univs_amd.ops on the GPU, the oracle stand-ins under cpu_ops)."""
    import torch.nn.functional as F
    dec = head.predictor
    T = g["g4_call0_output"].shape[1]
    mf = torch.from_numpy(g["g4_mask_features"])[0].to(device)                    # [T, C, H, W]
    targets = cases.targets_first_clip()
    # ---- G4: forward_prediction_heads (...decoder_univs.py:498-567)
    for k in (0, 4):
        tag = f"g4_call{k}_"
        out_tokens = torch.from_numpy(g[tag + "output"]).to(device)
        size = tuple(int(v) for v in g[tag + "target_size"])
        feat_lowres = ops.bilinear_resample(mf.contiguous(), size)
        cls_, msk_, attn_, _ = dec.forward_prediction_heads(out_tokens, mf.contiguous(), feat_lowres, "detection", targets, T,
                                                            need_masks=True)
        assert np.abs(cls_.cpu().numpy() - g[tag + "outputs_class"]).max() < 1e-4, tag
        ref_mask = g[tag + "outputs_mask"]
        assert np.abs(msk_.cpu().numpy() - ref_mask).max() < 1e-3, tag
        # the bool attention mask: the reference thresholds the bilinearly resized logits and resets all-True rows in the
        # layer loop (:390); ours does both in one operator.  A flip is tolerated only where the resized logit is within
        # 1e-4 of the threshold (there is none at these sizes on the CPU path).
        ref_attn = torch.from_numpy(g[tag + "attn_mask_head0"]).clone()            # [T, Q, hw]
        ref_attn[ref_attn.sum(-1) == ref_attn.shape[-1]] = False
        resized = F.interpolate(torch.from_numpy(ref_mask).flatten(0, 1), size=size, mode="bilinear", align_corners=False)
        resized = resized.permute(1, 0, 2, 3).flatten(2)                          # [T, Q, hw]
        diff = attn_.cpu() != ref_attn
        full_rows = (torch.from_numpy(g[tag + "attn_mask_head0"]).sum(-1) == ref_attn.shape[-1])
        near = resized.abs() < 1e-4
        assert not (diff & ~near & ~full_rows.unsqueeze(-1)).any(), (tag, int(diff.sum()))
    # ---- G5: one decoder layer, teacher-forced (:383-432)
    i = int(g["g5_layer"])
    output = torch.from_numpy(g["g5_output_in"]).to(device)
    src = torch.from_numpy(g["g5_src"]).to(device)
    pos = torch.from_numpy(g["g5_pos"]).to(device)
    qe = torch.from_numpy(g["g5_query_embed"]).to(device)
    attn_mask = torch.from_numpy(g["g5_attn_mask_head0"]).to(device)
    sa_mask = torch.from_numpy(g["g5_self_attn_mask"]).to(device)
    sa_mask = sa_mask if sa_mask.numel() else None
    Qn, bt, C = output.shape
    o = dec.transformer_cross_attention_layers[i](output, src.contiguous(), memory_mask=attn_mask, pos=None, query_pos=qe,
                                                  key=(src + pos).contiguous())
    o = dec.transformer_self_attention_layers[i](o.reshape(Qn * bt, 1, C), tgt_mask=sa_mask, query_pos=qe.reshape(Qn * bt, 1, C))
    o = dec.transformer_ffn_layers[i](o.reshape(Qn, bt, C))
    err = np.abs(o.cpu().numpy() - g["g5_output_out"]).max()
    assert err < 1e-4, ("g5", err)


def check_g2(pd, g, device, tol=2e-5):
    """ONE `MSDeformAttn.forward` (ops/modules/ms_deform_attn.py:82-121) and ONE encoder layer
    (msdeformattn.py:124-133) in isolation, against tensors captured with hooks inside the reference pixel decoder
    (oracle/gen_golden.py:g2_msdeformattn_layer).  `pd` is our MSDeformAttnPixelDecoder on `device`."""
    i = int(g["g2_layer"])
    layer = pd.transformer.encoder.layers[i]
    src = torch.from_numpy(g["g2_src"]).to(device)
    pos = torch.from_numpy(g["g2_pos"]).to(device)
    ref_pts = torch.from_numpy(g["g2_reference_points"]).to(device)
    shapes = [tuple(int(v) for v in hw) for hw in g["g2_spatial_shapes"]]
    lsi = [int(v) for v in g["g2_level_start_index"]]
    query = torch.from_numpy(g["g2_attn_query"]).to(device)
    assert np.abs((src + pos).cpu().numpy() - g["g2_attn_query"]).max() == 0
    attn = layer.self_attn(query, ref_pts, src, shapes, lsi, None)
    e1 = np.abs(attn.cpu().numpy() - g["g2_attn_out"]).max()
    out = layer(src, pos, ref_pts, shapes, lsi, None)
    e2 = np.abs(out.cpu().numpy() - g["g2_layer_out"]).max()
    assert e1 < tol and e2 < tol, (e1, e2)
    return e1, e2
