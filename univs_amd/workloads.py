"""Synthetic workloads of the BASELINE.json configurations: network geometries, closed-form inputs and builders of the
product modules with the closed-form weights (univs_amd/synth.py).  One definition shared by bench.py, tools/ and the
tests (tests/cases.py and tests/helpers.py import these names), and -- through oracle/gen_golden.py, which feeds the very
same tensors to the real reference -- by the golden fixtures.
"""
import torch

from . import synth
from .registry import ShapeSpec

# ---------------------------------------------------------------------------------------------------
# network geometries
# ---------------------------------------------------------------------------------------------------
SWIN_T = dict(pretrain_img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=[2, 2, 6, 2],
              num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
              ape=False, patch_norm=True)

R50_SHAPES = {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)}   # cfg 1
SWINT_SHAPES = {"res2": (96, 4), "res3": (192, 8), "res4": (384, 16), "res5": (768, 32)}

PIXDEC = dict(transformer_dropout=0.0, transformer_nheads=8, transformer_dim_feedforward=1024,
              transformer_enc_layers=6, conv_dim=256, mask_dim=256, norm="GN",
              transformer_in_features=["res3", "res4", "res5"], common_stride=4)
SWIN_B = dict(pretrain_img_size=384, patch_size=4, in_chans=3, embed_dim=128, depths=[2, 2, 18, 2],
              num_heads=[4, 8, 16, 32], window_size=12, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
              ape=False, patch_norm=True)
SWIN_L = dict(pretrain_img_size=384, patch_size=4, in_chans=3, embed_dim=192, depths=[2, 2, 18, 2],
              num_heads=[6, 12, 24, 48], window_size=12, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
              ape=False, patch_norm=True)
SWINL_SHAPES = {"res2": (192, 4), "res3": (384, 8), "res4": (768, 16), "res5": (1536, 32)}

# a small head case: features of a 64x96 padded input with R50 channel counts
HEAD_CASE = dict(name="head", T=2, H=64, W=96, Q=20, shapes=R50_SHAPES)
# BASELINE config 5's decoder length (T = 10 frames, 200 queries: a 2 000-token spatio-temporal self-attention, class / re-id
# means over 10 frames) on reduced-resolution features -- the decoder does not care about H x W, the reference's own 1080p run
# at T = 10 needs > 100 GB on the CPU (golden g6c)
HEAD_CASE_T10 = dict(name="head_t10", T=10, H=64, W=96, Q=200, shapes=R50_SHAPES)


def backbone_features(case=HEAD_CASE):
    """Synthetic res2..res5 (unit-variance, like LayerNorm'ed Swin outputs)."""
    feats = {}
    for k, (c, s) in case["shapes"].items():
        feats[k] = synth.normal(f"{case['name']}/feat/{k}", (case["T"], c, case["H"] // s, case["W"] // s))
    return feats


def clip_table():
    return synth.uniform("clip_cls_emb", (3938, 640))


def decoder_kwargs(case=HEAD_CASE, text_to_image=False, sa_mask="sep", num_dense_points=32, num_prev=5):
    return dict(in_channels=256, mask_classification=True, num_classes=133, hidden_dim=256,
                num_queries=case["Q"], nheads=8, dim_feedforward=2048, dec_layers=9, pre_norm=False,
                mask_dim=256, enforce_input_project=False, prompt_self_attn_layers=-1, num_frames=case["T"],
                num_dense_points=num_dense_points, text_prompt_enable=True, prompt_as_queries=True,
                text_prompt_to_image_enable=text_to_image, maskdec_self_attn_mask_type=sa_mask,
                position_embedding_sin3d_type="ArbitraryT", num_prev_frames_memory=num_prev,
                enabled_prev_frames_memory=True, enabled_prev_visual_prompts_for_grounding=False)


def sampler_kwargs(case=HEAD_CASE, num_dense_points=32, num_prev=5):
    return dict(pretrain_img_size=1024, hidden_dim=256, num_heads=8, num_frames=case["T"],
                num_prev_frames_memory=num_prev, num_dense_points=num_dense_points,
                position_embedding_sin3d_type="ArbitraryT", clip_stride=1)


def targets_first_clip(case=HEAD_CASE, task="detection", prompt_type="visual", dataset="ytvis_2021_dev"):
    return [{"task": task, "dataset_name": dataset, "prompt_type": prompt_type, "num_frames": case["T"],
             "first_frame_idx": 0, "frame_indices": torch.arange(0, case["T"])}]


def targets_with_entities(case=HEAD_CASE, first_frame_idx=1, n_ent=3):
    """Second clip of a video (stride 1): rectangular entity masks carried over from previous frames.
    masks/boxes cover the frames seen so far plus the (zero-padded) newest frame, as the clip loop leaves
    them (inference_video_entity.py:878-912)."""
    T, H, W = case["T"], case["H"], case["W"]
    t_hist = first_frame_idx + T          # frames 0 .. first_frame_idx+T-1
    masks = torch.zeros(n_ent, t_hist, H, W)
    boxes = torch.zeros(n_ent, t_hist, 4)
    for e in range(n_ent):
        for t in range(t_hist - 1):       # newest frame has no annotation yet
            y0, x0 = 4 + 9 * e + t, 6 + 17 * e + 2 * t
            hh, ww = 14 + 3 * e, 20 + 5 * e
            masks[e, t, y0:y0 + hh, x0:x0 + ww] = 1.0
            boxes[e, t] = torch.tensor([x0 / W, y0 / H, (x0 + ww) / W, (y0 + hh) / H])
    tv = targets_first_clip(case)[0]
    tv.update({"first_frame_idx": first_frame_idx,
               "frame_indices": torch.arange(first_frame_idx, first_frame_idx + T),
               "masks": masks, "boxes": boxes, "ids": torch.arange(n_ent)[:, None].repeat(1, t_hist),
               "first_appear_frame_idxs": torch.zeros(n_ent, dtype=torch.long)})
    return [tv]


# ---------------------------------------------------------------------------------------------------
# BASELINE config 2: Swin-T, T=5 @ 720p (padded 736x1280), 100 queries, first clip
# ---------------------------------------------------------------------------------------------------
CFG2 = dict(name="cfg2", T=5, H=720, W=1280, Q=100, shapes=SWINT_SHAPES)


def cfg2_frames(case=CFG2):
    return synth.synthetic_frames(case["T"], case["H"], case["W"], "frames/seed0")


# Swin-L (configs/univs_inf/vids/vis/univs_swinl_yt21_c1+univs.yaml:5-13) -- BASELINE config 5: T=10 @ 1080p (padded to
# 1088x1920), 200 queries.  The reference's CPU run of the full clip needs > 100 GB, so the golden (g19) is the same
# network on the first TWO frames; the T=10 run is checked through size-independent properties on the GPU.
# BASELINE config 4: Swin-B (window 12), T=5 @ 720p, 200 learnable queries + 4 referring expressions (grounding, 'sep-blocked'
# self-attention mask, text prompts fused into the image features: configs/univs_inf/vids/refvos/univs_swinb_refvos_davis_c1+univs.yaml)
SWINB_SHAPES = {"res2": (128, 4), "res3": (256, 8), "res4": (512, 16), "res5": (1024, 32)}
CFG4 = dict(name="cfg4", T=5, H=720, W=1280, Q=200, shapes=SWINB_SHAPES, n_exp=4)
CFG4_DECODER = dict(text_to_image=True, sa_mask="sep-blocked")


def targets_grounding(case=HEAD_CASE, n_exp=3):
    T = case["T"]
    tv = targets_first_clip(case, task="grounding", prompt_type="text")[0]
    tv["exp_word_feats"] = synth.normal("grounding/word", (n_exp, 77, T, 640))
    tv["exp_sentence_feats"] = synth.normal("grounding/sent", (n_exp, T, 640))
    tv["exp_word_len"] = [7] * n_exp
    return [tv]


def cfg4_targets(case=CFG4):
    return [targets_grounding(case, n_exp=case["n_exp"])[0]]


CFG5 = dict(name="cfg5", T=10, H=1080, W=1920, Q=200, shapes=SWINL_SHAPES)
CFG5_GOLDEN_T = 2


def cfg5_frames(T):
    return synth.synthetic_frames(T, CFG5["H"], CFG5["W"], "cfg5/frames")


def preprocess(frames, divisibility=32):
    """normalise + zero-pad to a multiple of 32 (univs/inference/inference_video_entity.py:251-260)."""
    mean = torch.tensor(synth.PIXEL_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(synth.PIXEL_STD).view(1, 3, 1, 1)
    x = (frames - mean) / std
    H, W = x.shape[-2:]
    Hp, Wp = (H + divisibility - 1) // divisibility * divisibility, (W + divisibility - 1) // divisibility * divisibility
    return torch.nn.functional.pad(x, (0, Wp - W, 0, Hp - H))


# ---------------------------------------------------------------------------------------------------
# product modules with the closed-form weights
# ---------------------------------------------------------------------------------------------------
def build_swin(device="cpu", variant=None, attn_mma="f16x3"):
    from .modeling.backbone.swin import SwinTransformer
    k = dict(variant or SWIN_T)
    m = SwinTransformer(k["pretrain_img_size"], k["patch_size"], k["in_chans"], k["embed_dim"], k["depths"],
                        k["num_heads"], k["window_size"], k["mlp_ratio"], k["qkv_bias"], k["qk_scale"], k["ape"],
                        k["patch_norm"]).eval()
    synth.load_synthetic(m, prefix="backbone.")
    return m.set_attention_mma(attn_mma).to(device)


def build_pixel_decoder(shapes, device="cpu"):
    from .modeling.pixel_decoder.msdeformattn import MSDeformAttnPixelDecoder
    ish = {k: ShapeSpec(channels=c, stride=s) for k, (c, s) in shapes.items()}
    m = MSDeformAttnPixelDecoder(ish, **PIXDEC).eval()
    synth.load_synthetic(m, prefix="sem_seg_head.pixel_decoder.")
    return m.to(device)


def build_head(case, device="cpu", return_aux=True, **dec_over):
    from .modeling.meta_arch.mask_former_head import MaskFormerHead
    from .modeling.prompt_encoder import VisualPromptSampler
    from .modeling.transformer_decoder.univs_decoder import VideoMultiScaleMaskedTransformerDecoderUniVS
    pd = build_pixel_decoder(case["shapes"])
    ish = {k: ShapeSpec(channels=c, stride=s) for k, (c, s) in case["shapes"].items()}
    dec = VideoMultiScaleMaskedTransformerDecoderUniVS(
        clip_class_embed_path=clip_table(), visual_prompt_sampler=VisualPromptSampler(**sampler_kwargs(case)),
        return_aux_outputs=return_aux, **decoder_kwargs(case, **dec_over)).eval()
    synth.load_synthetic(dec, prefix="sem_seg_head.predictor.")
    head = MaskFormerHead(ish, num_classes=133, pixel_decoder=pd, pixel_decoder_name="MSDeformAttnPixelDecoder",
                          transformer_predictor=dec, transformer_in_feature="multi_scale_pixel_decoder").eval()
    return head.to(device)


def build_model(device, case=None, variant=None, return_aux=False, **dec_over):
    """(backbone, head) of a workload on `device` with the closed-form weights: config 2 (Swin-T, 100 queries) by default."""
    case = CFG2 if case is None else case
    return build_swin(device, variant), build_head(case, device, return_aux=return_aux, **dec_over)
