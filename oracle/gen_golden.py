"""Generate the golden fixtures under tests/golden/ by running the REAL reference (imported from
/root/reference through oracle/ref_harness.py).  Dev-container only; the fixtures it writes are plain
data (inputs where they are not closed-form, and the reference's outputs) and travel with the repo.

    python oracle/gen_golden.py [name ...]        # no names = all

Inputs and weights come from univs_amd/synth.py (closed-form, name-keyed), so most fixtures only store
the reference's OUTPUTS; tests rebuild the inputs from the same names.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness as rh  # noqa: E402
from univs_amd import synth  # noqa: E402
from tests import cases  # noqa: E402  (shared input builders: the tests use the very same functions)

OUT = os.path.join(ROOT, "tests", "golden")
GENERATORS = {}


def gen(fn):
    GENERATORS[fn.__name__] = fn
    return fn


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ---------------------------------------------------------------------------------------------------
@gen
def g0_msda_kat():
    """The reference's own known-answer shapes (ops/test.py:24-31,35-63): N1 M2 D2 Lq2 L2 P2,
    shapes (6,4),(3,2), torch.manual_seed(3).  Inputs are stored (torch RNG is version dependent)."""
    R = rh.ref()
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    out = {}
    for tag in ("double", "float"):  # same call order as ops/test.py
        value = torch.rand(N, S, M, D) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        attn = torch.rand(N, Lq, M, L, P) + 1e-5
        attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
        if tag == "double":
            o = R.ms_deform_attn_core_pytorch(value.double(), shapes, loc.double(), attn.double())
        else:
            o = R.ms_deform_attn_core_pytorch(value, shapes, loc, attn)
        out.update({f"value_{tag}": value, f"loc_{tag}": loc, f"attn_{tag}": attn, f"out_{tag}": o})
    save("g0_msda_kat", shapes=shapes, level_start_index=lsi, **out)


@gen
def g1_msda_encoder_geometry():
    """MSDA at the config-1 encoder geometry incl. out-of-range locations;
    also a ragged / non-2x pyramid and an L=4 case.  Inputs closed-form (tests/cases.py)."""
    R = rh.ref()
    arrays = {}
    for case in cases.MSDA_CASES:
        value, shapes, lsi, loc, attn = cases.msda_inputs(case)
        o = R.ms_deform_attn_core_pytorch(value, torch.as_tensor(shapes), loc, attn)
        sub = cases.msda_query_subset(case, o.shape[1])
        arrays[f"{case['name']}/out_subset"] = o[:, sub]
        arrays[f"{case['name']}/sum"] = o.double().sum()
        arrays[f"{case['name']}/abs_sum"] = o.double().abs().sum()
    save("g1_msda_geometry", **arrays)


@gen
def g_window_attention():
    """Reference WindowAttention.forward (swin.py:131-171) with closed-form weights, with and without
    the shifted-window mask, window 7 (49 tokens) and window 12 (144 tokens)."""
    R = rh.ref()
    arrays = {}
    for case in cases.WINATTN_CASES:
        mod = R.WindowAttention(case["dim"], (case["win"], case["win"]), case["heads"]).eval()
        synth.load_synthetic(mod, prefix=case["name"] + ".")
        x, mask = cases.winattn_inputs(case)
        with torch.no_grad():
            y = mod(x, mask=mask)
        arrays[f"{case['name']}/out"] = y
    save("g_window_attention", **arrays)


# ---------------------------------------------------------------------------------------------------
# module-level goldens
# ---------------------------------------------------------------------------------------------------
def _ref_swin(R):
    m = R.SwinTransformer(drop_path_rate=0.3, **cases.SWIN_T)
    m.eval()  # the reference's SwinTransformer.train() returns None (swin.py:680-683): no chaining
    synth.load_synthetic(m, prefix="backbone.")
    return m


def _ref_pixel_decoder(R, shapes):
    ish = {k: R.ShapeSpec(channels=c, stride=s) for k, (c, s) in shapes.items()}
    m = R.MSDeformAttnPixelDecoder(ish, **cases.PIXDEC).eval()
    synth.load_synthetic(m, prefix="sem_seg_head.pixel_decoder.")
    return m, ish


def _ref_head(R, case, **dec_over):
    pd, ish = _ref_pixel_decoder(R, case["shapes"])
    clip_path = "/tmp/univs_clip_cls_emb.pth"
    torch.save(cases.clip_table(), clip_path)
    sampler = R.VisualPromptSampler(**cases.sampler_kwargs(case))
    dec = R.Decoder(clip_class_embed_path=clip_path, visual_prompt_sampler=sampler,
                    **cases.decoder_kwargs(case, **dec_over)).eval()
    synth.load_synthetic(dec, prefix="sem_seg_head.predictor.")
    head = R.MaskFormerHead(ish, num_classes=133, pixel_decoder=pd, pixel_decoder_name="MSDeformAttnPixelDecoder",
                            transformer_predictor=dec, transformer_in_feature="multi_scale_pixel_decoder").eval()
    return head


def _tensor_fields(tv):
    return {k: v for k, v in tv.items() if isinstance(v, torch.Tensor)}


@gen
def g_state_dict_layout():
    """Parameter names + shapes of the reference module tree (checkpoint compatibility)."""
    R = rh.ref()
    layout = {}
    for k, v in _ref_swin(R).state_dict().items():
        layout["backbone." + k] = list(v.shape)
    head = _ref_head(R, cases.HEAD_CASE, text_to_image=True)
    for k, v in head.state_dict().items():
        layout["sem_seg_head." + k] = list(v.shape)
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "state_dict_layout.json"), "w") as f:
        json.dump({"note": "Swin-T backbone + MaskFormerHead(R50-channel pixel decoder, UniVS decoder Q=20, "
                           "text_prompt_to_image_enable=True)", "layout": layout}, f, indent=0)
    print("wrote state_dict_layout.json", len(layout), "entries")


@gen
def g9_swin():
    R = rh.ref()
    out = _ref_swin(R)(cases.swin_input())
    save("g9_swin", **out)


@gen
def g10_position_embeddings():
    R = rh.ref()
    x = torch.zeros(2, 8, 5, 7)
    arrays = {"sine2d": R.PositionEmbeddingSine(4, normalize=True)(x)}
    x5 = torch.zeros(1, 3, 8, 5, 7)
    fi = torch.tensor([[4, 5, 9]])
    pts = synth.uniform("pe/pts", (6, 2), 0.0, 1.0)
    arb = R.PositionEmbeddingSine3DArbitraryT(4, normalize=True)
    fix = R.PositionEmbeddingSine3D(4, normalize=True)
    arrays["arb3d"] = arb(x5, fi)
    arrays["arb3d_default_t"] = arb(x5)
    arrays["arb_points"] = arb.forward_points_with_size((3, 40, 56), pts, 7)
    arrays["arb_points_vec"] = arb.forward_points_with_size((3, 40, 56), pts, torch.tensor([2, 3, 4]))
    arrays["fix3d"] = fix(x5)
    arrays["fix_points"] = fix.forward_points_with_size((3, 40, 56), pts)
    save("g10_position_embeddings", **arrays)


@gen
def g3_pixel_decoder():
    R = rh.ref()
    pd, _ = _ref_pixel_decoder(R, cases.HEAD_CASE["shapes"])
    mf, mf_bfe, enc0, ms = pd.forward_features(cases.backbone_features())
    save("g3_pixel_decoder", mask_features=mf, mask_features_bfe_conv=mf_bfe, enc0=enc0,
         ms0=ms[0], ms1=ms[1], ms2=ms[2])


def _head_outputs(out):
    d = {k: out[k] for k in ("pred_logits", "pred_masks", "pred_embds")}
    if isinstance(out.get("pred_reid_logits"), torch.Tensor):
        d["pred_reid_logits"] = out["pred_reid_logits"]
    # every layer's mask logits as well: lets tests tell "wrong" from "an attention-mask bit flipped"
    for i, a in enumerate(out["aux_outputs"]):
        d[f"aux{i}_pred_masks"] = a["pred_masks"]
    return d


@gen
def g6_head_first_clip():
    R = rh.ref()
    head = _ref_head(R, cases.HEAD_CASE)
    out = head(cases.backbone_features(), targets=cases.targets_first_clip())
    save("g6_head_first_clip", **_head_outputs(out))


@gen
def g6c_head_t10_q200():
    """the decoder at BASELINE config 5's length -- T = 10 frames, 200 learnable queries (2 000-token
    spatio-temporal self-attention, class logits averaged over 10 frames) -- through the unmodified reference head on
    reduced-resolution synthetic features (64 x 96: the decoder is agnostic to H x W; the reference's own 1080p run at T = 10
    needs > 100 GB).  Stored: every 4th query of the mask logits / embeddings (all frames); every 16th class logit, the largest class
    logit and its index per query; per-query magnitude and positive-pixel count of the mask logits."""
    R = rh.ref()
    case = cases.HEAD_CASE_T10
    head = _ref_head(R, case)
    out = head(cases.backbone_features(case), targets=cases.targets_first_clip(case))
    save("g6c_head_t10_q200", pred_logits_k16=out["pred_logits"][:, :, ::16], pred_logits_max=out["pred_logits"].amax(-1),
         pred_logits_argmax=out["pred_logits"].argmax(-1), pred_masks_q4=out["pred_masks"][:, ::4],
         pred_embds_q4=out["pred_embds"][:, ::4], pred_masks_absmax=out["pred_masks"].abs().amax(dim=(0, 2, 3, 4)),
         pred_masks_positive=(out["pred_masks"] > 0).sum(dim=(0, 2, 3, 4)))


@gen
def g4_g5_teacher_forced():
    """per-call fixtures from INSIDE the reference decoder (first clip of HEAD_CASE),
    captured with module hooks while the unmodified reference runs:
      G4  forward_prediction_heads (...decoder_univs.py:498-567): inputs (query states, mask features, target size) and
          outputs (class logits, mask logits, bool attention mask of head 0 -- the reference repeats it over the heads) of
          the first call (learnable queries only) and of the call after layer 3;
      G5  decoder layer 4 teacher-forced (:383-432): query states before the layer, query embedding, the level's memory /
          position tensors, the attention mask actually used (after the all-True-row reset of :390), the self-attention
          mask; and the query states after the FFN.
    The mask threshold (sigmoid < 0.5) makes the end-to-end head discontinuous; these pin the pieces around it."""
    R = rh.ref()
    head = _ref_head(R, cases.HEAD_CASE)
    dec = head.predictor
    d = {}
    calls = {"n": 0}
    orig_heads = dec.forward_prediction_heads

    def heads_hook(output, mask_features, attn_mask_target_size, task, targets):
        res = orig_heads(output, mask_features, attn_mask_target_size=attn_mask_target_size, task=task, targets=targets)
        k = calls["n"]
        calls["n"] += 1
        if k in (0, 4):
            tag = f"g4_call{k}_"
            nh = dec.num_heads
            d[tag + "output"] = output.detach().clone()
            d[tag + "target_size"] = torch.tensor(list(attn_mask_target_size))
            d[tag + "outputs_class"] = res[0].detach().clone()
            d[tag + "outputs_mask"] = res[1].detach().clone()
            d[tag + "attn_mask_head0"] = res[2].detach()[0::nh].clone()      # [(b t) h, q, hw] -> head 0 of every frame
            if k == 0:
                d["g4_mask_features"] = mask_features.detach().clone()
        return res

    dec.forward_prediction_heads = heads_hook
    LAYER = 4
    nh = dec.num_heads

    def cross_pre(mod, args, kwargs):
        d["g5_output_in"] = args[0].detach().clone()
        d["g5_src"] = args[1].detach().clone()
        d["g5_attn_mask_head0"] = kwargs["memory_mask"].detach()[0::nh].clone()
        d["g5_pos"] = kwargs["pos"].detach().clone()
        d["g5_query_embed"] = kwargs["query_pos"].detach().clone()

    def self_pre(mod, args, kwargs):
        m = kwargs.get("tgt_mask")
        d["g5_self_attn_mask"] = m.detach().clone() if m is not None else torch.zeros(0)

    def ffn_post(mod, args, out):
        d["g5_output_out"] = out.detach().clone()

    h1 = dec.transformer_cross_attention_layers[LAYER].register_forward_pre_hook(cross_pre, with_kwargs=True)
    h2 = dec.transformer_self_attention_layers[LAYER].register_forward_pre_hook(self_pre, with_kwargs=True)
    h3 = dec.transformer_ffn_layers[LAYER].register_forward_hook(ffn_post)
    head(cases.backbone_features(), targets=cases.targets_first_clip())
    for h in (h1, h2, h3):
        h.remove()
    d["g5_layer"] = torch.tensor(LAYER)
    save("g4_g5_teacher_forced", **d)


@gen
def g7_head_visual_prompts():
    """Second clip with visual prompts: creates the memory pool.  torch.manual_seed(0) right before the
    call fixes the sampler's randperm draws (prompt_encoder.py:420,424,481)."""
    R = rh.ref()
    head = _ref_head(R, cases.HEAD_CASE)
    targets = cases.targets_with_entities()
    torch.manual_seed(0)
    out = head(cases.backbone_features(), targets=targets)
    d = _head_outputs(out)
    for k in ("prompt_feats", "prompt_pe", "prompt_attn_masks", "prompt_obj_ids"):
        d["pool_" + k] = targets[0][k]
    # third clip on the same dict: exercises zero_pad_prompt + pool update + memory read
    tv = targets[0]
    T = cases.HEAD_CASE["T"]
    tv["first_frame_idx"] = 2
    tv["frame_indices"] = torch.arange(2, 2 + T)
    tv["masks"] = torch.cat([tv["masks"], torch.zeros_like(tv["masks"][:, :1])], 1)
    tv["masks"][:, -2] = tv["masks"][:, -3]
    tv["boxes"] = torch.cat([tv["boxes"], torch.zeros_like(tv["boxes"][:, :1])], 1)
    tv["boxes"][:, -2] = tv["boxes"][:, -3]
    tv["ids"] = torch.cat([tv["ids"], tv["ids"][:, :1]], 1)
    torch.manual_seed(1)
    out3 = head(cases.backbone_features(), targets=targets)
    for k, v in _head_outputs(out3).items():
        if not k.startswith("aux"):
            d["clip3_" + k] = v
    for k in ("prompt_feats", "prompt_pe", "prompt_attn_masks"):
        d["clip3_pool_" + k] = targets[0][k]
    save("g7_head_visual_prompts", **d)


@gen
def g8_head_grounding():
    R = rh.ref()
    head = _ref_head(R, cases.HEAD_CASE, text_to_image=True, sa_mask="sep-blocked")
    out = head(cases.backbone_features(), targets=cases.targets_grounding())
    save("g8_head_grounding", **_head_outputs(out))


@gen
def g8b_head_detection_text():
    """Category-guided path (task 'detection', prompt_type 'text', dataset 'vspw': 124 class prompts)."""
    R = rh.ref()
    head = _ref_head(R, cases.HEAD_CASE, text_to_image=True)
    out = head(cases.backbone_features(), targets=cases.targets_first_clip(prompt_type="text", dataset="vspw"))
    d = _head_outputs(out)
    d = {k: v for k, v in d.items() if not k.startswith("aux")}   # 144 queries: keep the fixture small
    save("g8b_head_detection_text", **d)


@gen
def g12_cfg2_full_size():
    """BASELINE config 2 through the real reference on CPU (about half a minute): strided samples and
    checksums of every stage, so that full-size GPU runs can be compared with the reference itself."""
    import hashlib
    R = rh.ref()
    case = cases.CFG2
    swin = _ref_swin(R)
    head = _ref_head(R, case)
    x = cases.preprocess(cases.cfg2_frames())
    feats = swin(x)
    out = head(feats, targets=cases.targets_first_clip(case))
    d = {}
    for k, v in feats.items():
        d["feat_" + k + "_s"] = v[:, ::8, ::4, ::4]
        d["feat_" + k + "_abs_mean"] = v.double().abs().mean()
    pm = out["pred_masks"]
    d["pred_masks_s"] = pm[0, :, :, ::16, ::16]
    d["pred_masks_abs_mean"] = pm.double().abs().mean()
    d["pred_masks_pos_count"] = (pm > 0).sum()
    d["pred_masks_near_zero_1e-3"] = (pm.abs() < 1e-3).sum()
    d["pred_masks_sign_sha256"] = np.frombuffer(
        hashlib.sha256(np.packbits((pm > 0).numpy()).tobytes()).digest(), dtype=np.uint8)
    d["pred_logits"] = out["pred_logits"]
    d["pred_embds"] = out["pred_embds"]
    for i in (0, 4, 8):   # a few intermediate layers: tells "wrong" from "an attention-mask bit flipped"
        d[f"aux{i}_pred_masks_s"] = out["aux_outputs"][i]["pred_masks"][0, :, :, ::16, ::16]
    save("g12_cfg2_full_size", **d)


@gen
def g9b_swin_b():
    """Swin-B (window 12): exercises the 144-token window-attention kernel inside the full backbone."""
    R = rh.ref()
    m = R.SwinTransformer(drop_path_rate=0.3, **cases.SWIN_B)
    m.eval()
    synth.load_synthetic(m, prefix="backbone.")
    out = m(cases.swin_input(cases.SWINB_CASE))
    save("g9b_swin_b", **{k: v[:, ::2] for k, v in out.items()})   # every 2nd channel keeps the fixture small


# ---------------------------------------------------------------------------------------------------
# G11: the clip loop (SURVEY.md section 8c "clip loop counterpart"): the REFERENCE's
# InferenceVideoEntity.inference_video on a 7-frame synthetic video, `targets[0]` dumped at the entry of
# every head call (= the state the previous clip left behind) and at the end.
# ---------------------------------------------------------------------------------------------------
LOOP_STATE_KEYS = ("logits", "masks", "mask_logits", "boxes", "embds", "ids", "first_appear_frame_idxs",
                   "mask_quality_scores", "occurrence", "prompt_pe", "prompt_feats", "prompt_attn_masks",
                   "frame_indices")


def _capture_sampler_draws(enc, store):
    """Record, inside the unmodified reference, WHICH pixels every `get_mask_prompt` call (prompt_encoder.py:168-263) samples:
    the point chosen by `select_points_from_box_mask` (:420,424: `randperm` over the entity's candidate pixels) as a flat
    index y * w + x of the full-resolution mask, and the `num_dense_points` feature-map pixels chosen by
    `get_dense_features` (:471-481: all pixels cyclically for small masks, `randperm(count)[:R]` otherwise) as flat indices
    (-1 = empty mask).  The draw SIZES are pixel counts, so a fp32 implementation whose mask differs in one near-threshold
    pixel cannot reproduce the draws from the seed; replaying the sampled pixels removes that discontinuity."""
    if hasattr(enc, "draw_log"):
        # the build's encoder (tests/test_b1_dropin_cpu.py drives OUR modules with the reference's loop): it samples all key
        # frames of a clip in one pass and keeps its own record in the same format
        enc.draw_log = store
        return
    orig_gmp, orig_sel, orig_gdf = enc.get_mask_prompt, enc.select_points_from_box_mask, enc.get_dense_features
    state = {}

    def sel(h_img, w_img, boxes=None, masks=None, **k):
        pc = orig_sel(h_img, w_img, boxes=boxes, masks=masks, **k)
        if masks is not None:
            h, w = masks.shape[-2:]
            x = torch.round(pc[:, 0] * w - 0.5).long()
            y = torch.round(pc[:, 1] * h - 0.5).long()
            assert ((x + 0.5) / w - pc[:, 0]).abs().max() < 1e-6 and ((y + 0.5) / h - pc[:, 1]).abs().max() < 1e-6
            state["point_idx"] = (y * w + x).to(torch.int32)
        return pc

    def gdf(img_features, img_pos, masks_binary, query_pe, query_feats, **k):
        perms = []
        orig_rp = torch.randperm

        def rp(n, *a, **kw):
            p_ = orig_rp(n, *a, **kw)
            perms.append(p_.clone())
            return p_
        torch.randperm = rp
        try:
            out = orig_gdf(img_features, img_pos, masks_binary, query_pe, query_feats, **k)
        finally:
            torch.randperm = orig_rp
        R = enc.num_dense_points
        it = iter(perms)
        rows = []
        for m in masks_binary:
            fi = torch.nonzero(m.flatten()).reshape(-1)
            if len(fi) == 0:
                rows.append(torch.full((R,), -1, dtype=torch.long))
            elif len(fi) < R:
                rows.append(fi.repeat(int(R / len(fi)) + 1)[:R])
            else:
                rows.append(fi[next(it)[:R]])
        fidx = torch.stack(rows)
        feats = img_features.flatten(-2).t()
        ok = fidx[:, 0] >= 0
        assert torch.equal(out[0][ok][:, :, 0], feats[fidx[ok]]), "restated dense-token indices do not reproduce the reference"
        state["feat_idx"] = fidx.to(torch.int32)
        return out

    def gmp(*a, **k):
        state.clear()
        out = orig_gmp(*a, **k)
        store.append((state["point_idx"].clone(), state["feat_idx"].clone()))
        return out
    enc.get_mask_prompt, enc.select_points_from_box_mask, enc.get_dense_features = gmp, sel, gdf


def _ref_loop(case, model, **over):
    RI = rh.ref_inference()
    kw = cases.loop_kwargs(case, **over)
    kw.update(overlap_threshold=0.0, metadata=None, LSJ_aug_image_size=1024, LSJ_aug_enable_test=False,
              sem_seg_postprocess_before_inference=False, num_classes=133, data_name="ytvis_2021_dev",
              prompt_as_queries=True, zero_shot_inference=False, semantic_on=False, instance_on=True,
              panoptic_on=False, tracker_type="", window_inference=False, is_multi_cls=True, merge_on_cpu=False,
              num_max_inst_test=50, output_dir="/tmp")
    inf = RI.InferenceVideoEntity(**kw)
    inf.save_results_vis = lambda *a, **k: []                  # result formats are out of scope (needs pycocotools)
    RI.module.vis_clip_instances_to_coco_json_video = lambda bi, res, **k: res
    dumps = {}
    head = model.sem_seg_head
    calls = []

    def snapshot(tag, tv):
        for k in LOOP_STATE_KEYS:
            if k in tv:
                if case.get("reduce"):
                    dumps.update(cases.loop_reduce(tag, k, tv[k]))
                else:
                    dumps[f"{tag}_{k}"] = tv[k].detach().clone().float() if tv[k].dtype == torch.bool else tv[k].detach().clone()

    draws, draws_clip = [], []
    pred = getattr(head, "predictor", None)
    if pred is not None and getattr(pred, "visual_prompt_sampler", None) is not None:
        _capture_sampler_draws(pred.visual_prompt_sampler.visual_prompt_encoder, draws)

    def hooked(features, targets=None, **k):
        snapshot(f"clip{len(calls)}_in", targets[0])
        calls.append(int(targets[0]["first_frame_idx"]))
        print("      clip at frame", calls[-1], file=sys.stderr, flush=True)
        n0 = len(draws)
        res = head(features, targets=targets, **k)
        draws_clip.extend([len(calls) - 1] * (len(draws) - n0))
        return res
    model = types.SimpleNamespace(backbone=model.backbone, sem_seg_head=hooked)
    x = cases.preprocess(cases.loop_frames(case))
    images = types.SimpleNamespace(tensor=x, image_sizes=[case["image_size"]] * case["n_frames"])
    targets = cases.loop_targets(case)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):            # the reference prints shapes per new entity (:843)
        torch.manual_seed(1)
        inf.inference_video(model, cases.loop_batched_inputs(case), images, targets)
    snapshot("final", targets[0])
    dumps["clip_first_frames"] = torch.tensor(calls)
    if draws:
        # the sampler's draws of every get_mask_prompt call, in call order (see _capture_sampler_draws)
        dumps["draws_n"] = torch.tensor([len(p_) for p_, _ in draws], dtype=torch.int32)
        dumps["draws_clip"] = torch.tensor(draws_clip, dtype=torch.int32)
        dumps["draws_point_idx"] = torch.cat([p_ for p_, _ in draws])
        dumps["draws_feat_idx"] = torch.cat([f_ for _, f_ in draws])
    return dumps


@gen
def g11a_clip_loop_model():
    """real backbone + head (synthetic weights): windowed backbone, stride rule, memory pool across 3 clips"""
    R = rh.ref()
    case = cases.LOOP_CASE
    model = types.SimpleNamespace(backbone=_ref_swin(R), sem_seg_head=_ref_head(R, case))
    d = _ref_loop(case, model, stability_score_thresh=0.0)
    print("   clips at", d["clip_first_frames"].tolist(), "entities", d["final_ids"].tolist())
    save("g11a_clip_loop_model", **d)


@gen
def g20_cfg3_long_video():
    """BASELINE config 3: the reference's sliding clip loop (inference_video_entity.py:301-404) over a 40-frame 720p video,
    Swin-T, 100 queries, clips of 5 frames at the reference's default stride 1 (MODEL.UniVS.TEST.CLIP_STRIDE; strides 2, 4 and 5 make the reference's own
    memory-pool update raise at T=5, stride 3 does on this video from a later clip on), windows of 5 frames; reduced per-clip states."""
    R = rh.ref()
    case = cases.CFG3_LOOP
    model = types.SimpleNamespace(backbone=_ref_swin(R), sem_seg_head=_ref_head(R, case))
    d = _ref_loop(case, model, stability_score_thresh=0.0, clip_stride=1)
    print("   clips at", d["clip_first_frames"].tolist(), "entities", d["final_ids"].tolist())
    save("g20_cfg3_long_video", **d)


@gen
def g19_cfg5_swinl_1080p():
    """BASELINE config 5's network (Swin-L window 12 + head, 200 queries) at 1080p on the first TWO frames of the clip
    through the real reference on CPU: strided samples + checksums, as G12 / G14."""
    import hashlib
    R = rh.ref()
    case = dict(cases.CFG5, T=cases.CFG5_GOLDEN_T)
    swin = R.SwinTransformer(drop_path_rate=0.3, **cases.SWIN_L)
    swin.eval()
    synth.load_synthetic(swin, prefix="backbone.")
    head = _ref_head(R, case)
    x = cases.preprocess(cases.cfg5_frames(case["T"]))
    feats = swin(x)
    out = head(feats, targets=cases.targets_first_clip(case))
    d = {}
    for k, v in feats.items():
        d["feat_" + k + "_s"] = v[:, ::16, ::4, ::4]
    pm = out["pred_masks"]
    d["pred_masks_s"] = pm[0, :, :, ::16, ::16]
    d["pred_masks_abs_mean"] = pm.double().abs().mean()
    d["pred_masks_pos_count"] = (pm > 0).sum()
    d["pred_masks_near_zero_1e-3"] = (pm.abs() < 1e-3).sum()
    d["pred_masks_sign_sha256"] = np.frombuffer(
        hashlib.sha256(np.packbits((pm > 0).numpy()).tobytes()).digest(), dtype=np.uint8)
    d["pred_logits"] = out["pred_logits"]
    d["pred_embds"] = out["pred_embds"][:, :, :, ::4]
    save("g19_cfg5_swinl_1080p", **d)


@gen
def g19b_cfg5_reference_autocast():
    """A yardstick for BASELINE config 5's fp16 variant: how far does the REFERENCE sit from its own fp32 run when it is evaluated
    the way train_net.py:334 evaluates it (`with autocast(): inference_on_dataset(...)`: fp16 Linears / matmuls in the backbone
    and the decoder, the pixel decoder pinned to fp32 by `@autocast(enabled=False)`, msdeformattn.py:316)?  Emulated on the CPU:
    torch.autocast("cpu", dtype=float16) around backbone + head, the pixel decoder's forward_features taken out of the region as
    the reference's decorator does on CUDA.  The CPU and CUDA autocast op lists differ slightly (both lower Linear / conv / matmul /
    bmm and keep softmax / layer_norm in fp32), so this is an emulation, stated as such in the test.  Same input, weights and
    strided samples as g19; stores the autocast run's deviations from the fp32 run of this process (scalars only)."""
    R = rh.ref()
    case = dict(cases.CFG5, T=cases.CFG5_GOLDEN_T)
    swin = R.SwinTransformer(drop_path_rate=0.3, **cases.SWIN_L)
    swin.eval()
    synth.load_synthetic(swin, prefix="backbone.")
    head = _ref_head(R, case)
    x = cases.preprocess(cases.cfg5_frames(case["T"]))
    pd_forward = head.pixel_decoder.forward_features

    def fp32_pixel_decoder(features, *a, **k):
        with torch.autocast("cpu", enabled=False):
            return pd_forward({n: f.float() for n, f in features.items()}, *a, **k)
    head.pixel_decoder.forward_features = fp32_pixel_decoder

    def run():
        feats = swin(x)
        out = head(feats, targets=cases.targets_first_clip(case))
        return ({k: v.float()[:, ::16, ::4, ::4] for k, v in feats.items()}, out["pred_masks"].float(), out["pred_logits"].float())
    f32, pm32, pl32 = run()
    with torch.autocast("cpu", dtype=torch.float16):
        f16, pm16, pl16 = run()
    d = {"emulation": "torch.autocast('cpu', dtype=float16) around backbone + head; pixel decoder outside the region"}
    for k in f32:
        d["feat_" + k + "_err"] = (f16[k] - f32[k]).abs().max()
    d["pred_masks_err_s"] = (pm16 - pm32)[0, :, :, ::16, ::16].abs().max()
    d["pred_masks_err_full"] = (pm16 - pm32).abs().max()
    d["pred_masks_abs_max"] = pm32.abs().max()
    d["pred_masks_sign_flips"] = ((pm16 > 0) != (pm32 > 0)).sum()
    d["pred_masks_sign_flips_beyond_5e-3"] = (((pm16 > 0) != (pm32 > 0)) & (pm32.abs() > 5e-3)).sum()
    d["pred_logits_err"] = (pl16 - pl32).abs().max()
    print("   reference under emulated autocast vs its fp32 run:", {k: float(v) for k, v in d.items() if k != "emulation"})
    save("g19b_cfg5_reference_autocast", **d)


@gen
def g11b_clip_loop_scripted():
    """scripted scene (tests/cases.py ScriptedHead): several entities, NMS, a newcomer, a leaver"""
    case = cases.SCRIPT_CASE
    model = types.SimpleNamespace(backbone=cases.ScriptedBackbone(), sem_seg_head=cases.ScriptedHead())
    d = _ref_loop(case, model)
    print("   clips at", d["clip_first_frames"].tolist(), "entities", d["final_ids"].tolist(),
          "first seen", d["final_first_appear_frame_idxs"].tolist())
    save("g11b_clip_loop_scripted", **d)


# edge cases of the clip schedule on the scripted scene: a video shorter than a clip, exactly one clip, stride 1 with the
# backbone window equal to the clip length
LOOP_EDGE_CASES = {
    "g11e_loop_short_video": (dict(n_frames=2), {}),
    "g11f_loop_one_clip": (dict(n_frames=3), dict(clip_stride=1)),
    "g11g_loop_stride1_window3": (dict(n_frames=5), dict(clip_stride=1, num_frames_window_test=3)),
}


def _loop_edge(name):
    case_over, kw_over = LOOP_EDGE_CASES[name]
    case = dict(cases.SCRIPT_CASE, **case_over)
    model = types.SimpleNamespace(backbone=cases.ScriptedBackbone(), sem_seg_head=cases.ScriptedHead(case))
    d = _ref_loop(case, model, **kw_over)
    print("   clips at", d["clip_first_frames"].tolist(), "entities", d["final_ids"].tolist() if "final_ids" in d else None)
    save(name, **d)


@gen
def g11e_loop_short_video():
    _loop_edge("g11e_loop_short_video")


@gen
def g11f_loop_one_clip():
    _loop_edge("g11f_loop_one_clip")


@gen
def g11g_loop_stride1_window3():
    _loop_edge("g11g_loop_stride1_window3")


def _loop_scene(scene):
    case = cases.SCRIPT_CASE
    model = types.SimpleNamespace(backbone=cases.ScriptedBackbone(),
                                  sem_seg_head=cases.ScriptedHead(case, cases.SCRIPT_SCENES[scene]))
    d = _ref_loop(case, model)
    print("   clips at", d["clip_first_frames"].tolist(), "entities", d["final_ids"].tolist() if "final_ids" in d else None,
          "keys", len(d))
    save(f"g11_scene_{scene}", **d)


@gen
def g11h_scene_empty():
    _loop_scene("empty")


@gen
def g11i_scene_late():
    _loop_scene("late")


@gen
def g11j_scene_leavers():
    _loop_scene("leavers")


@gen
def g13_msda_backward():
    """gradients of the REFERENCE's ms_deform_attn_core_pytorch by autograd (float64 -> stored as float32)"""
    R = rh.ref()
    case = cases.MSDA_BWD_CASE
    value, shapes, lsi, loc, attn = cases.msda_inputs(case)
    go = synth.normal("msda_bwd/go/" + case["name"], (value.shape[0], loc.shape[1], value.shape[2] * value.shape[3]))
    with torch.enable_grad():
        v = value.double().requires_grad_(True)
        l_ = loc.double().requires_grad_(True)
        a = attn.double().requires_grad_(True)
        out = R.ms_deform_attn_core_pytorch(v, torch.as_tensor(shapes), l_, a)
        out.backward(go.double())
    save("g13_msda_backward", out=out.detach().float(), grad_value=v.grad.float(), grad_sampling_loc=l_.grad.float(),
         grad_attn_weight=a.grad.float())


@gen
def g14_cfg4_full_size():
    """BASELINE config 4 (Swin-B, grounding with 4 expressions, 200 queries, T=5 @ 720p) through the real
    reference on CPU: strided samples + checksums, as G12."""
    import hashlib
    R = rh.ref()
    case = cases.CFG4
    swin = R.SwinTransformer(drop_path_rate=0.3, **cases.SWIN_B)
    swin.eval()
    synth.load_synthetic(swin, prefix="backbone.")
    head = _ref_head(R, case, **cases.CFG4_DECODER)
    x = cases.preprocess(cases.cfg2_frames())
    feats = swin(x)
    out = head(feats, targets=cases.cfg4_targets(case))
    d = {}
    for k, v in feats.items():
        d["feat_" + k + "_s"] = v[:, ::16, ::4, ::4]
    pm = out["pred_masks"]
    d["pred_masks_s"] = pm[0, :, :, ::16, ::16]
    d["pred_masks_abs_mean"] = pm.double().abs().mean()
    d["pred_masks_pos_count"] = (pm > 0).sum()
    d["pred_masks_near_zero_1e-3"] = (pm.abs() < 1e-3).sum()
    d["pred_masks_sign_sha256"] = np.frombuffer(
        hashlib.sha256(np.packbits((pm > 0).numpy()).tobytes()).digest(), dtype=np.uint8)
    d["pred_logits"] = out["pred_logits"]
    d["pred_embds"] = out["pred_embds"][:, :, :, ::4]
    if out.get("pred_reid_logits") is not None and isinstance(out["pred_reid_logits"], torch.Tensor):
        d["pred_reid_logits"] = out["pred_reid_logits"]
    save("g14_cfg4_full_size", **d)


@gen
def g2_msdeformattn_layer():
    """ONE `MSDeformAttn.forward` (ops/modules/ms_deform_attn.py:82-121) and ONE encoder layer
    (msdeformattn.py:124-133) in isolation, captured with hooks on encoder layer 2 of the reference pixel decoder while
    it runs on HEAD_CASE's backbone features: inputs (src, pos, reference points, level table) and outputs."""
    R = rh.ref()
    pd, _ = _ref_pixel_decoder(R, cases.HEAD_CASE["shapes"])
    LAYER = 2
    layer = pd.transformer.encoder.layers[LAYER]
    d = {}

    def layer_pre(mod, args, kwargs):
        names = ("src", "pos", "reference_points", "spatial_shapes", "level_start_index")
        for k, v in list(zip(names, args)) + list(kwargs.items()):
            if isinstance(v, torch.Tensor) and k in names:
                d["g2_" + k] = v.detach().clone()

    def attn_pre(mod, args, kwargs):
        d["g2_attn_query"] = args[0].detach().clone()

    def attn_post(mod, args, out):
        d["g2_attn_out"] = out.detach().clone()

    def layer_post(mod, args, out):
        d["g2_layer_out"] = out.detach().clone()

    hs = [layer.register_forward_pre_hook(layer_pre, with_kwargs=True), layer.register_forward_hook(layer_post),
          layer.self_attn.register_forward_pre_hook(attn_pre, with_kwargs=True), layer.self_attn.register_forward_hook(attn_post)]
    pd.forward_features(cases.backbone_features())
    for h in hs:
        h.remove()
    d["g2_layer"] = torch.tensor(LAYER)
    assert {"g2_src", "g2_pos", "g2_reference_points", "g2_attn_out", "g2_layer_out"} <= set(d), sorted(d)
    save("g2_msdeformattn_layer", **d)


CFG4_NEAR_EPS = 1e-2      # |resized mask logit| below this is "near the attention-mask threshold"


@gen
def g14b_cfg4_attn_masks():
    """Teacher-forcing data for BASELINE config 4 (the run of g14): inside the unmodified reference decoder,
      * the bool attention mask every decoder layer actually used -- `memory_mask` of its cross-attention AFTER the
        all-True-row reset (...decoder_univs.py:390, :400-405), head 0 of every frame (the reference repeats it over the 8
        heads, :565), bit-packed along the key axis;
      * for every prediction-head call, the entries whose resized mask logit (:555-566, before `sigmoid < 0.5`) lies
        within CFG4_NEAR_EPS of the threshold: flat index into [T, Q', HW_l] and the reference's value.  A free-running
        fp32 implementation may put exactly these entries on the other side; the GPU test names them.
    Also the outputs again (same reduction as g14) so that the file is self-contained."""
    import torch.nn.functional as F
    R = rh.ref()
    case = cases.CFG4
    swin = R.SwinTransformer(drop_path_rate=0.3, **cases.SWIN_B)
    swin.eval()
    synth.load_synthetic(swin, prefix="backbone.")
    head = _ref_head(R, case, **cases.CFG4_DECODER)
    dec = head.predictor
    nh = dec.num_heads
    d = {}
    calls = {"n": 0}
    orig_heads = dec.forward_prediction_heads

    def heads_hook(output, mask_features, attn_mask_target_size, task, targets):
        res = orig_heads(output, mask_features, attn_mask_target_size=attn_mask_target_size, task=task, targets=targets)
        k = calls["n"]
        calls["n"] += 1
        om = res[1]                                                     # [b, q, t, H, W]
        b = om.shape[0]
        lr = F.interpolate(om.flatten(0, 1), size=attn_mask_target_size, mode="bilinear", align_corners=False)
        lr = lr.view(b, om.shape[1], om.shape[2], -1).permute(0, 2, 1, 3).flatten(0, 1)      # [(b t), q, hw]
        assert torch.equal(lr.sigmoid() < 0.5, res[2][0::nh]), "restated resize does not reproduce the reference's mask"
        near = (lr.abs() < CFG4_NEAR_EPS).flatten().nonzero().flatten()
        d[f"call{k}_near_idx"] = near.to(torch.int32)
        d[f"call{k}_near_val"] = lr.flatten()[near].clone()
        d[f"call{k}_shape"] = torch.tensor(list(lr.shape))
        print(f"      heads call {k}: target {tuple(attn_mask_target_size)}, {near.numel()} entries within {CFG4_NEAR_EPS}", file=sys.stderr)
        return res

    dec.forward_prediction_heads = heads_hook
    hooks = []
    for i, layer in enumerate(dec.transformer_cross_attention_layers):
        def cross_pre(mod, args, kwargs, i=i):
            m = kwargs["memory_mask"].detach()[0::nh]                   # [T, Q', HW_l] bool, rows already reset
            d[f"layer{i}_attn_mask_bits"] = torch.from_numpy(np.packbits(m.numpy(), axis=-1))
            d[f"layer{i}_attn_mask_shape"] = torch.tensor(list(m.shape))
        hooks.append(layer.register_forward_pre_hook(cross_pre, with_kwargs=True))
    x = cases.preprocess(cases.cfg2_frames())
    out = head(swin(x), targets=cases.cfg4_targets(case))
    for h in hooks:
        h.remove()
    pm = out["pred_masks"]
    d["pred_masks_s"] = pm[0, :, :, ::16, ::16]
    d["near_eps"] = torch.tensor(CFG4_NEAR_EPS)
    save("g14b_cfg4_attn_masks", **d)


@gen
def g11c_clip_loop_vss():
    """semantic sub-task ('vss': non-overlapping clips, per-clip class x mask maps) on the scripted scene"""
    RI = rh.ref_inference()
    case = cases.SCRIPT_CASE
    kw = cases.loop_kwargs(case)
    kw.update(overlap_threshold=0.0, metadata=None, LSJ_aug_image_size=1024, LSJ_aug_enable_test=False,
              sem_seg_postprocess_before_inference=False, num_classes=133, data_name="vspw_vss_video_dev",
              prompt_as_queries=True, zero_shot_inference=False, semantic_on=True, instance_on=False,
              panoptic_on=False, tracker_type="", window_inference=False, is_multi_cls=True, merge_on_cpu=False,
              num_max_inst_test=50, output_dir="/tmp")
    inf = RI.InferenceVideoEntity(**kw)
    calls = []
    head = cases.ScriptedHead(case)

    def hooked(features, targets=None, **k):
        calls.append(int(targets[0]["first_frame_idx"]))
        return head(features, targets=targets, **k)
    model = types.SimpleNamespace(backbone=cases.ScriptedBackbone(), sem_seg_head=hooked)
    x = cases.preprocess(cases.loop_frames(case))
    images = types.SimpleNamespace(tensor=x, image_sizes=[case["image_size"]] * case["n_frames"])
    targets = cases.loop_targets(case)
    targets[0]["sub_task"] = "vss"
    targets[0]["dataset_name"] = "vspw_vss_video_dev"
    res = inf.inference_video(model, cases.loop_batched_inputs(case), images, targets)
    print("   clips at", calls, "sem map", tuple(res["pred_masks"].shape), "classes", res["pred_masks"].unique().tolist())
    save("g11c_clip_loop_vss", pred_masks=res["pred_masks"], clip_first_frames=torch.tensor(calls))


@gen
def g11d_clip_loop_vps():
    """panoptic sub-task ('vps') on the scripted scene: thing / stuff de-duplication, segment-id memory"""
    RI = rh.ref_inference()
    case = cases.SCRIPT_CASE
    kw = cases.loop_kwargs(case)
    meta = types.SimpleNamespace(thing_dataset_id_to_contiguous_id={c: i for i, c in enumerate(cases.SCRIPT_THING_IDS)})
    kw.update(overlap_threshold=0.4, metadata=meta, LSJ_aug_image_size=1024, LSJ_aug_enable_test=False,
              sem_seg_postprocess_before_inference=False, num_classes=133, data_name="vipseg_panoptic_val",
              prompt_as_queries=True, zero_shot_inference=False, semantic_on=False, instance_on=False,
              panoptic_on=True, tracker_type="", window_inference=False, is_multi_cls=True, merge_on_cpu=False,
              num_max_inst_test=50, output_dir="/tmp")
    inf = RI.InferenceVideoEntity(**kw)
    dumps, calls = {}, []
    head = cases.ScriptedHead()

    def hooked(features, targets=None, **k):
        tv = targets[0]
        for key in LOOP_STATE_KEYS:
            if key in tv:
                v = tv[key].detach().clone()
                dumps[f"clip{len(calls)}_in_{key}"] = v.float() if v.dtype == torch.bool else v
        calls.append(int(tv["first_frame_idx"]))
        return head(features, targets=targets, **k)
    model = types.SimpleNamespace(backbone=cases.ScriptedBackbone(), sem_seg_head=hooked)
    x = cases.preprocess(cases.loop_frames(case))
    images = types.SimpleNamespace(tensor=x, image_sizes=[case["image_size"]] * case["n_frames"])
    targets = cases.loop_targets(case)
    targets[0].update(sub_task="vps", dataset_name="vipseg_panoptic_val")
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        res = inf.inference_video(model, cases.loop_batched_inputs(case), images, targets)
    infos = sorted((d["id"], int(d["isthing"]), d["category_id"]) for d in res["segments_infos"])
    print("   clips at", calls, "panoptic", tuple(res["pred_masks"].shape), "segments", infos)
    save("g11d_clip_loop_vps", pred_masks=res["pred_masks"], segments_infos=np.array(infos, dtype=np.int64),
         clip_first_frames=torch.tensor(calls), **dumps)


# ---------------------------------------------------------------------------------------------------
# G15: the VOS / RefVOS drivers (univs/inference/inference_video_vos.py) on the scripted scene
# ---------------------------------------------------------------------------------------------------
VOS_STATE_KEYS = ("masks", "mask_logits", "boxes", "embds", "labels", "first_appear_frame_idxs", "frame_indices")


class _RefAnn:
    """stand-in for detectron2 Instances as the reference's VOS driver uses it (:586-607)"""

    def __init__(self, image_size, ori_ids, gt_masks, gt_boxes, gt_classes):
        self.image_size, self.ori_ids, self.gt_masks, self.gt_classes = tuple(image_size), list(ori_ids), gt_masks, gt_classes
        self.gt_boxes = types.SimpleNamespace(tensor=gt_boxes)

    def __len__(self):
        return len(self.ori_ids)

    def to(self, device):
        return self


def _ref_vos(mode, targets, tag, case=None, objects=None, **kw_over):
    import glob
    import shutil
    from PIL import Image
    RV = rh.ref_inference_vos()
    case = case or cases.SCRIPT_CASE
    out_dir = f"/tmp/univs_vos_{tag}"
    shutil.rmtree(out_dir, ignore_errors=True)
    kw = cases.vos_kwargs(case, video_unified_inference_queries=mode)
    kw.update(object_mask_threshold=0.8, overlap_threshold=0.8, overlap_threshold_entity=0.5, stability_score_thresh=0.0,
              metadata=None, LSJ_aug_image_size=1024, LSJ_aug_enable_test=False, sem_seg_postprocess_before_inference=False,
              num_classes=133, data_name="ytbvos18_val", zero_shot_inference=False, semantic_on=False, instance_on=True,
              panoptic_on=False, test_topk_per_image=100, tracker_type="", window_inference=False, output_dir=out_dir)
    kw.update(kw_over)
    inf = RV.InferenceVideoVOS(**kw)
    dumps, calls = {}, []
    head = cases.ScriptedHead(case, objects) if objects is not None else cases.ScriptedHead(case)

    def hooked(features, targets=None, **k):
        calls.append(int(targets[0]["first_frame_idx"]))
        return head(features, targets=targets, **k)
    orig_write = inf.write_predictions_into_annotations_per_clip

    def write_and_dump(out, image_size, tg, first_frame_idx, stride):
        orig_write(out, image_size, tg, first_frame_idx, stride)
        for key in VOS_STATE_KEYS:
            v = tg[0][key].detach().clone()
            dumps[f"clip{len(calls) - 1}_out_{key}"] = v.float() if v.dtype == torch.bool else v
    inf.write_predictions_into_annotations_per_clip = write_and_dump
    model = types.SimpleNamespace(backbone=cases.ScriptedBackbone(), sem_seg_head=hooked)
    x = cases.preprocess(cases.loop_frames(case))
    images = types.SimpleNamespace(tensor=x, image_sizes=[case["image_size"]] * case["n_frames"])
    inf.inference_video_vos(model, cases.loop_batched_inputs(case), images, targets, case["image_size"], case["image_size"])
    dumps["clip_first_frames"] = torch.tensor(calls)
    # read the PNG results back
    ann = os.path.join(out_dir, "inference/Annotations", "clip0")
    if targets[0]["task"] == "sot":
        files = sorted(glob.glob(ann + "/*.png"))
        dumps["result_idmaps"] = torch.from_numpy(np.stack([np.array(Image.open(f)) for f in files]))
        dumps["result_frames"] = torch.tensor([int(os.path.basename(f)[:5]) for f in files])
    else:
        for eid in sorted(os.listdir(ann)):
            files = sorted(glob.glob(os.path.join(ann, eid, "*.png")))
            dumps[f"result_exp{eid}"] = torch.from_numpy(np.stack([np.array(Image.open(f)) for f in files]))
            dumps[f"result_exp{eid}_frames"] = torch.tensor([int(os.path.basename(f)[:5]) for f in files])
    shutil.rmtree(out_dir, ignore_errors=True)
    return dumps


@gen
def g15a_vos_prompt():
    d = _ref_vos("prompt", cases.vos_targets_sot(_RefAnn), "a")
    print("   clips", d["clip_first_frames"].tolist(), "frames written", d["result_frames"].tolist(),
          "ids", d["result_idmaps"].unique().tolist())
    save("g15a_vos_prompt", **d)


@gen
def g15b_vos_prompt_learn():
    d = _ref_vos("prompt+learn", cases.vos_targets_sot(_RefAnn), "b")
    print("   clips", d["clip_first_frames"].tolist(), "ids", d["result_idmaps"].unique().tolist())
    save("g15b_vos_prompt_learn", **d)


@gen
def g15d_vos_learn():
    d = _ref_vos("learn", cases.vos_targets_sot(_RefAnn), "d")
    print("   clips", d["clip_first_frames"].tolist(), "ids", d["result_idmaps"].unique().tolist())
    save("g15d_vos_learn", **d)


@gen
def g15e_vos_short_tail():
    """4-frame video, stride 2: the second clip is the 2-frame tail; object 33 is annotated in the LAST frame"""
    case = dict(cases.SCRIPT_CASE, n_frames=4)
    d = _ref_vos("prompt", cases.vos_targets_sot(_RefAnn, case), "e", case=case)
    print("   clips", d["clip_first_frames"].tolist(), "frames", d["result_frames"].tolist(), "ids", d["result_idmaps"].unique().tolist())
    save("g15e_vos_short_tail", **d)


@gen
def g15f_vos_stride1():
    """5-frame video, stride 1 (every clip finishes one frame), prompt + learnable queries"""
    case = dict(cases.SCRIPT_CASE, n_frames=5)
    d = _ref_vos("prompt+learn", cases.vos_targets_sot(_RefAnn, case), "f", case=case, clip_stride=1)
    print("   clips", d["clip_first_frames"].tolist(), "frames", d["result_frames"].tolist(), "ids", d["result_idmaps"].unique().tolist())
    save("g15f_vos_stride1", **d)


def _viposeg(mode, tag):
    case = cases.VIPOSEG_CASE
    targets = cases.vos_targets_sot(_RefAnn, case, objects=cases.VIPOSEG_OBJECTS, class_offset=cases.VIPOSEG_CLASS_START,
                                    dataset="viposeg_val")
    meta = types.SimpleNamespace(stuff_dataset_id_to_contiguous_id={i: i - 1 for i in cases.VIPOSEG_STUFF_IDS})
    d = _ref_vos(mode, targets, tag, case=case, objects=cases.VIPOSEG_OBJECTS, metadata=meta, data_name="viposeg_val")
    print("   clips", d["clip_first_frames"].tolist(), "ids", d["result_idmaps"].unique().tolist())
    return d


@gen
def g15g_viposeg_prompt():
    save("g15g_viposeg_prompt", **_viposeg("prompt", "g"))


@gen
def g15h_viposeg_prompt_learn():
    save("g15h_viposeg_prompt_learn", **_viposeg("prompt+learn", "h"))


@gen
def g15c_rvos_grounding():
    d = _ref_vos("prompt", cases.vos_targets_grounding(), "c")
    print("   clips", d["clip_first_frames"].tolist(), [k for k in d if k.startswith("result_exp") and not k.endswith("frames")])
    save("g15c_rvos_grounding", **d)


# ---------------------------------------------------------------------------------------------------
# G16: the CLIP text encoder + tokenizer that feed the grounding prompts (univs/modeling/language/)
# ---------------------------------------------------------------------------------------------------
TEXT_EXPRESSIONS = ["a dog running", "The person in a red-and-white jacket, skiing downhill!", "two zebras' heads (left)",
                    "naive cafe - 3 cats & 12 dogs", "it's the giraffe that's eating leaves",
                    "antidisestablishmentarianism supercalifragilistic", "a " * 100]
TEXT_SMALL, TEXT_FULL = cases.TEXT_SMALL, cases.TEXT_FULL      # TEXT_FULL: RN50x4 text tower (TextEncoder.py:157-174)


def _ref_text_encoder(L, cfg):
    m = L.TextEncoder.CLIPLangEncoder(out_features=["res5"], freeze_at=0, **cfg)
    return synth.load_synthetic(m, "lang_encoder.").eval()


@gen
def g16a_tokenizer():
    L = rh.ref_language()
    U = L.clip_prompt_utils
    ids = U.pre_tokenize_expression(TEXT_EXPRESSIONS)
    cls = U.pre_tokenize(["person", "traffic light", ["tv", "television"]][:2])
    tok = U.SimpleTokenizer()
    plain = U.tokenize(TEXT_EXPRESSIONS[:6])
    cleaned = U.clean_strings(["Traffic_light(1)", "a man's hat - red/blue!"])
    save("g16a_tokenizer", expressions=np.array(TEXT_EXPRESSIONS), ids=ids.to(torch.int32), class_ids=cls.to(torch.int32),
         plain=plain.to(torch.int32), cleaned=np.array(cleaned),
         decoded=np.array([tok.decode(tok.encode(e)) for e in TEXT_EXPRESSIONS[:6]]))


def _text_case(L, cfg, n_exp):
    enc = _ref_text_encoder(L, cfg)
    tokens = L.clip_prompt_utils.pre_tokenize_expression(TEXT_EXPRESSIONS[:n_exp])       # [E, 81, 77]
    sample = tokens[:, :3].reshape(-1, 77)                                                  # a few texts
    with torch.no_grad():
        x_word, x_eot = enc.encode_text(sample, only_eot=False)
        only = enc.encode_text(sample, only_eot=True)
    assert torch.equal(only, x_eot)
    tpe = L.TextPromptEncoder(enc, num_frames=2)
    with torch.no_grad():
        w, s_, n = tpe.get_expression_prompt(TEXT_EXPRESSIONS[:n_exp], torch.device("cpu"))
    return dict(tokens=tokens.to(torch.int32), sample=sample.to(torch.int32), x_word=x_word, x_eot=x_eot,
                exp_word_feats=w[:, :, 0], exp_sentence_feats=s_[:, 0], exp_word_len=np.array(n))


@gen
def g16b_text_encoder_small():
    d = _text_case(rh.ref_language(), TEXT_SMALL, 4)
    print("   x_eot", tuple(d["x_eot"].shape), float(d["x_eot"].abs().max()), "word feats", tuple(d["exp_word_feats"].shape))
    save("g16b_text_encoder_small", **d)


@gen
def g16c_text_encoder_full():
    d = _text_case(rh.ref_language(), TEXT_FULL, 2)
    d["x_word"] = d["x_word"][:, ::4]                 # keep the fixture small: every 4th token
    d["exp_word_feats"] = d["exp_word_feats"][:, ::4]
    print("   x_eot", tuple(d["x_eot"].shape), float(d["x_eot"].abs().max()))
    save("g16c_text_encoder_full", **d)


# ---------------------------------------------------------------------------------------------------
# G17: result merging of the VIS loop (univs/inference/comm.py:97-207).  pycocotools is absent: its `encode` is
# backed by OUR run-length coder here, so this golden pins the score / merge / top-k logic, not the RLE strings.
# ---------------------------------------------------------------------------------------------------
@gen
def g17_vis_results():
    from univs_amd.inference import results as ours
    RI = rh.ref_inference()
    mu = sys.modules["pycocotools.mask"]

    def encode(arr):            # [H, W, 1] Fortran uint8 -> list of RLE dicts with bytes counts, as pycocotools returns
        r = ours.rle_encode_masks(torch.from_numpy(np.ascontiguousarray(arr[:, :, 0])).bool())[0]
        return [{"size": r["size"], "counts": r["counts"].encode("ascii")}]
    mu.encode = encode
    out = {}
    for tag, kw in (("default", {}), ("tight", dict(apply_cls_thresh=0.5, test_topk_per_video=2))):
        info, clips = cases.vis_result_records()
        for clip in clips:
            for r in clip:
                m = r.pop("masks")
                r["segmentations"] = [dict(x) for x in ours.rle_encode_masks(m)]
        res = RI.comm.vis_clip_instances_to_coco_json_video(info, clips, **kw)
        out[f"{tag}_score"] = np.array([r["score"] for r in res])
        out[f"{tag}_category"] = np.array([r["category_id"] for r in res])
        out[f"{tag}_areas"] = np.array([[ours.rle_area(s) for s in r["segmentations"]] for r in res])
        out[f"{tag}_video_id"] = np.array([r["video_id"] for r in res])
        print("  ", tag, len(res), "records")
    sc = torch.stack([cases.vis_result_records()[1][c][1]["score"] for c in range(3)]).clone()
    out["consistency"] = RI.comm.calculate_mask_temporal_consistency_scores(sc.clone())
    save("g17_vis_results", **out)


# ---------------------------------------------------------------------------------------------------
# G18: caller-side target preparation (univs/prepare_targets.py: process_inference)
# ---------------------------------------------------------------------------------------------------
def _describe_targets(out, prefix):
    """dict list -> flat {name: array}: tensors as they are, everything else as one json string per video"""
    d = {}
    for i, tv in enumerate(out):
        plain = {}
        for k, v in tv.items():
            if isinstance(v, torch.Tensor) and v.numel() > 100_000:      # e.g. the class-embedding table: sample + checksum
                d[f"{prefix}_{i}_{k}_shape"] = np.array(v.shape)
                d[f"{prefix}_{i}_{k}_sample"] = v[::97, ::16]
                d[f"{prefix}_{i}_{k}_sum"] = v.double().sum()
            elif isinstance(v, torch.Tensor):
                d[f"{prefix}_{i}_{k}"] = v
            else:
                plain[k] = v
        d[f"{prefix}_{i}_plain"] = np.array(json.dumps(plain, sort_keys=True, default=list))
    return d


@gen
def g18_prepare_targets():
    PT = rh.ref_prepare_targets()
    L = rh.ref_language()
    table_path = "/tmp/univs_clip_table.pth"
    torch.save(cases.clip_table(), table_path)
    enc = _ref_text_encoder(L, TEXT_SMALL)
    tpe = L.TextPromptEncoder(enc, num_frames=3)
    d = {}
    for name, (over, batched) in cases.prepare_targets_inputs().items():
        pt = PT(num_frames=3, clip_class_embed_path=table_path, **over)
        with torch.no_grad():
            out = pt.process_inference(batched, (64, 96), torch.device("cpu"), tpe, (60, 90))
        d.update(_describe_targets(out, name))
        d[f"{name}_input_prompt_type"] = np.array(batched[0]["prompt_type"])
        print("  ", name, sorted(out[0].keys()))
    save("g18_prepare_targets", **d)


# ---------------------------------------------------------------------------------------------------
# G21: on-disk VPS / VSS result formats (univs/evaluation/vps_evaluation.py:117-178, vss_evaluation.py:93-118)
# ---------------------------------------------------------------------------------------------------
@gen
def g21_result_files():
    import tempfile
    from PIL import Image
    E = rh.ref_evaluators()
    out = {}
    with tempfile.TemporaryDirectory() as d:
        gt = os.path.join(d, "gt.json")
        with open(gt, "w") as f:
            json.dump({"categories": list(cases.VPS_CATEGORIES.values())}, f)
        ev = object.__new__(E.VPSEvaluator)
        ev._metadata = types.SimpleNamespace(categories=cases.VPS_CATEGORIES)
        ev._output_dir, ev.pan_gt_json_file, ev._predictions = os.path.join(d, "vps"), gt, []
        os.makedirs(os.path.join(ev._output_dir, "pan_pred"), exist_ok=True)
        inputs, outputs = cases.result_file_inputs(), cases.vps_result_outputs()
        np.random.seed(7)                         # IdGenerator draws the second colour of a thing class from numpy's global generator
        ev.process([inputs], outputs)
        names = sorted(os.listdir(os.path.join(ev._output_dir, "pan_pred", "vid_0007")))
        out["vps_png_names"] = np.array(json.dumps(names))
        out["vps_png"] = np.stack([np.asarray(Image.open(os.path.join(ev._output_dir, "pan_pred", "vid_0007", n))) for n in names])
        out["vps_record"] = np.array(json.dumps(ev._predictions[0], sort_keys=True, default=int))
        ev._distributed, ev._do_evaluation = False, False
        ev._logger = types.SimpleNamespace(warning=print)
        ev.evaluate()
        out["vps_pred_json"] = np.array(open(os.path.join(ev._output_dir, "pred.json")).read())
        es = object.__new__(E.VSSEvaluator)
        es._output_dir, es.ignore_val = os.path.join(d, "vss"), 255
        es.contiguous_id_to_dataset_id = dict(cases.VSS_CONTIGUOUS_TO_DATASET)
        es.process([inputs], cases.vss_result_outputs())
        names = sorted(os.listdir(os.path.join(es._output_dir, "vid_0007")))
        out["vss_png_names"] = np.array(json.dumps(names))
        out["vss_png"] = np.stack([np.asarray(Image.open(os.path.join(es._output_dir, "vid_0007", n))) for n in names])
    print("  ", out["vps_png"].shape, out["vss_png"].shape, len(str(out["vps_record"])))
    save("g21_result_files", **out)


def main():
    names = sys.argv[1:] or list(GENERATORS)
    for n in names:
        print(f"== {n}")
        with torch.no_grad():
            GENERATORS[n]()


if __name__ == "__main__":
    main()
