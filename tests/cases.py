"""Closed-form test inputs shared by oracle/gen_golden.py (which feeds them to the real reference) and
by the tests (which feed the very same tensors to the oracle and to the HIP path)."""
import math

import numpy as np
import torch

from univs_amd import synth

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


def maskdec_inputs(case):
    e = synth.normal(case["name"] + "/embed", (case["T"], case["Q"], case["C"]), std=0.5)
    f = synth.normal(case["name"] + "/feat", (case["T"], case["C"], case["H"], case["W"]), std=0.5)
    return e, f


# ---------------------------------------------------------------------------------------------------
# Module-level cases (Swin, pixel decoder, UniVS decoder / head)
# ---------------------------------------------------------------------------------------------------
SWIN_T = dict(pretrain_img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=[2, 2, 6, 2],
              num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
              ape=False, patch_norm=True)
SWIN_CASE = dict(name="swin_t", N=2, H=96, W=128)

R50_SHAPES = {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)}   # cfg 1
SWINT_SHAPES = {"res2": (96, 4), "res3": (192, 8), "res4": (384, 16), "res5": (768, 32)}

PIXDEC = dict(transformer_dropout=0.0, transformer_nheads=8, transformer_dim_feedforward=1024,
              transformer_enc_layers=6, conv_dim=256, mask_dim=256, norm="GN",
              transformer_in_features=["res3", "res4", "res5"], common_stride=4)

HEAD_CASE = dict(name="head", T=2, H=64, W=96, Q=20, shapes=R50_SHAPES)   # features for a 64x96 padded input


def clip_table():
    return synth.uniform("clip_cls_emb", (3938, 640))


def swin_input(case=SWIN_CASE):
    frames = synth.synthetic_frames(case["N"], case["H"], case["W"], "swin/frames")
    mean = torch.tensor(synth.PIXEL_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(synth.PIXEL_STD).view(1, 3, 1, 1)
    return (frames - mean) / std


def backbone_features(case=HEAD_CASE):
    """Synthetic res2..res5 (unit-variance, like LayerNorm'ed Swin outputs)."""
    feats = {}
    for k, (c, s) in case["shapes"].items():
        feats[k] = synth.normal(f"{case['name']}/feat/{k}", (case["T"], c, case["H"] // s, case["W"] // s))
    return feats


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


def targets_grounding(case=HEAD_CASE, n_exp=3):
    T = case["T"]
    tv = targets_first_clip(case, task="grounding", prompt_type="text")[0]
    tv["exp_word_feats"] = synth.normal("grounding/word", (n_exp, 77, T, 640))
    tv["exp_sentence_feats"] = synth.normal("grounding/sent", (n_exp, T, 640))
    tv["exp_word_len"] = [7] * n_exp
    return [tv]


# ---------------------------------------------------------------------------------------------------
# BASELINE config 2: Swin-T, T=5 @ 720p (padded 736x1280), 100 queries, first clip
# ---------------------------------------------------------------------------------------------------
CFG2 = dict(name="cfg2", T=5, H=720, W=1280, Q=100, shapes=SWINT_SHAPES)


def cfg2_frames(case=CFG2):
    return synth.synthetic_frames(case["T"], case["H"], case["W"], "frames/seed0")


def preprocess(frames, divisibility=32):
    """normalise + zero-pad to a multiple of 32 (univs/inference/inference_video_entity.py:251-260)."""
    mean = torch.tensor(synth.PIXEL_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(synth.PIXEL_STD).view(1, 3, 1, 1)
    x = (frames - mean) / std
    H, W = x.shape[-2:]
    Hp, Wp = (H + divisibility - 1) // divisibility * divisibility, (W + divisibility - 1) // divisibility * divisibility
    return torch.nn.functional.pad(x, (0, Wp - W, 0, Hp - H))


# Swin-B (configs/univs_inf/vids/refvos/univs_swinb_refvos_davis_c1+univs.yaml:5-9): window 12 -> 144-token
# windows, the second instantiation of the window-attention kernel
SWIN_B = dict(pretrain_img_size=384, patch_size=4, in_chans=3, embed_dim=128, depths=[2, 2, 18, 2],
              num_heads=[4, 8, 16, 32], window_size=12, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
              ape=False, patch_norm=True)
SWINB_CASE = dict(name="swin_b", N=1, H=96, W=160)
