"""GPU: the product modules with the real HIP operators against (a) the golden outputs of the REAL
reference (tests/golden/) at the small scale, (b) the reference's own config-2 (720p, T=5, Q=100) outputs
(g12: strided samples + checksums), (c) the CPU oracle path on the same seeded inputs."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle.cpu_path import cpu_ops
from tests import cases, helpers
from univs_amd import layers, ops

pytestmark = pytest.mark.gpu


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _to(d, dev):
    return {k: v.to(dev) for k, v in d.items()}


def _targets_to(targets, dev):
    out = []
    for tv in targets:
        out.append({k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in tv.items()})
    return out


# Which Linears may leave the hand-written three-product kernels for the library GEMM, per BASELINE config: (what, K, N) as
# univs_amd.layers counts them (LIBRARY_LINEAR_COUNTS).  A hot Linear that quietly falls through fails the config's test.
# (the decoder FFN's second Linear, 2048 -> 256 on Q' T rows, nine layers: K = 2048 is wider than the few-rows kernel takes)
_FFN2 = {("F.linear", 2048, 256)}
LIBRARY_LINEARS_ALLOWED = {"cfg2": _FFN2, "cfg4": _FFN2, "cfg5": _FFN2}
# queries of config 4 that own an attention-mask entry on the other side of the threshold: 13 in round 6 (gpurun_out/r06_n; the
# reference logits of those entries are all below 1.4e-5 in magnitude) -- profiles/r06_cfg4_flipped_queries_v1.txt
CFG4_MAX_FLIPPED_QUERIES = 20


def _check_library_linears(tag, allowed):
    got = dict(layers.LIBRARY_LINEAR_COUNTS)
    print(f"{tag}: Linears on the library GEMM (what, K, N) -> calls: {got}")
    if allowed is not None:
        extra = {k: v for k, v in got.items() if k not in allowed}
        assert not extra, f"{tag}: Linears left the hand-written kernels: {extra}"


def test_swin_matches_reference(cuda, golden_dir):
    g = _g(golden_dir, "g9_swin")
    swin = helpers.build_swin(cuda)
    with torch.no_grad():
        out = swin(cases.swin_input().to(cuda))
    for k in ("res2", "res3", "res4", "res5"):
        err = np.abs(out[k].cpu().numpy() - g[k]).max()
        assert err < 5e-4, (k, err)


def test_pixel_decoder_matches_reference(cuda, golden_dir):
    g = _g(golden_dir, "g3_pixel_decoder")
    pd = helpers.build_pixel_decoder(cases.HEAD_CASE["shapes"], cuda)
    with torch.no_grad():
        mf, mf_bfe, enc0, ms = pd.forward_features(_to(cases.backbone_features(), cuda))
    assert ops.msda_last_impl() == 2, "an LDS-tiled MSDA kernel must be the one that ran"
    got = dict(mask_features=mf, mask_features_bfe_conv=mf_bfe, enc0=enc0, ms0=ms[0], ms1=ms[1], ms2=ms[2])
    for k, v in got.items():
        err = np.abs(v.cpu().numpy() - g[k]).max()
        assert err < 5e-4, (k, err)


@pytest.mark.parametrize("name,dec_over,targets_fn,seed", helpers.HEAD_SCENARIOS, ids=[s[0] for s in helpers.HEAD_SCENARIOS])
def test_head_matches_reference(cuda, golden_dir, name, dec_over, targets_fn, seed):
    """north-star contract: mask logits within 1e-3 max-abs of the reference's CPU path, sign-identical."""
    g = _g(golden_dir, name)
    head = helpers.build_head(cases.HEAD_CASE, cuda, **dec_over)
    targets = _targets_to(targets_fn(), cuda)
    with torch.no_grad():
        if seed is not None:
            torch.manual_seed(seed)
        out = head(_to(cases.backbone_features(), cuda), targets=targets)
    helpers.check_head_outputs(out, g, "", tol=1e-3)
    if name == "g7_head_visual_prompts":
        for k in ("prompt_feats", "prompt_pe"):
            assert np.abs(targets[0][k].cpu().numpy() - g["pool_" + k]).max() < 1e-3, k
        assert (targets[0]["prompt_attn_masks"].cpu().numpy() == g["pool_prompt_attn_masks"]).all()
        helpers.advance_to_third_clip(targets)
        with torch.no_grad():
            torch.manual_seed(1)
            out3 = head(_to(cases.backbone_features(), cuda), targets=targets)
        helpers.check_head_outputs(out3, g, "clip3_", tol=1e-3)


def test_device_sampler_draws_are_reproducible_on_the_gpu(cuda):
    """The device-side prompt sampler on the GPU: after `begin_video` under the same `torch.manual_seed`, the same inputs give the same
    points and dense tokens bit for bit, whatever was drawn in between; another seed gives other draws; the draws lie inside the masks."""
    from tests.test_sampler_device_cpu import HF, R, S, WF, encoders, scene
    dev = encoders()[1]
    masks, feats, pos = (t.to(cuda) for t in scene())
    runs = []
    for seed in (5, 5, 6):
        torch.manual_seed(seed)
        dev.begin_video(cuda)
        runs.append(dev.get_mask_prompt(feats, pos, masks, key_fid=0, key_fid_original=5))
        dev.select_points_from_box_mask(HF, WF, masks=masks, num_points=17)          # other draws in between
    assert all(torch.equal(a, b) for a, b in zip(runs[0], runs[1]))
    assert not torch.equal(runs[0][2], runs[2][2])
    p, pd, fd, am = runs[0]
    fm = torch.nn.functional.interpolate(masks[:, None], (HF, WF), mode="nearest")[:, 0] >= 0.5
    for e in (0, 3):                                                          # R distinct pixels of the feature mask (channel 0 = pixel index)
        idx = fd[e, :, 0, 0].long()
        assert fm[e].flatten()[idx].all() and idx.unique().numel() == R
    assert fd[2].abs().max() == 0                                             # the empty entity


def test_fused_proca_equals_the_layered_path(cuda):
    """ProCA without building `memory` (univs_decoder._proca_fused: q / k0 / v0 in one few-rows launch, the dense tokens' K / V
    Linears on the tokens in place, ops.proca_attention, out_proj + residual + LayerNorm in one launch) == the reference's layered
    sequence (concatenations, nn.MultiheadAttention) on the second clip of the visual-prompt scenario; and it is the path that runs."""
    from univs_amd.switches import override
    name, dec_over, targets_fn, seed = [s for s in helpers.HEAD_SCENARIOS if s[0] == "g7_head_visual_prompts"][0]
    head = helpers.build_head(cases.HEAD_CASE, cuda, **dec_over)
    calls = []
    orig = ops.proca_attention
    ops.proca_attention = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        outs = []
        for fused in (True, False):
            n0 = len(calls)
            with torch.no_grad(), override(fused_proca=fused):
                torch.manual_seed(seed)
                outs.append(head(_to(cases.backbone_features(), cuda), targets=_targets_to(targets_fn(), cuda)))
            assert (len(calls) - n0 > 0) == fused
    finally:
        ops.proca_attention = orig
    assert len(calls) == len(head.predictor.transformer_prompt_self_attention_layers)     # one attention launch per ProCA layer
    for k in ("pred_masks", "pred_logits", "pred_embds"):
        err = (outs[0][k] - outs[1][k]).abs().max().item()
        assert err < 2e-4 * max(1.0, outs[1][k].abs().max().item()), (k, err)
    assert ((outs[0]["pred_masks"] > 0) != (outs[1]["pred_masks"] > 0))[outs[1]["pred_masks"].abs() > 1e-3].sum() == 0


def test_prompted_clip_aten_operator_budget(cuda):
    """The steady-state clip of a video (second clip, 10 entities carried as visual prompts, BASELINE config 2's size) is launch-bound on
    the host: the number of ATen operators that launch a kernel -- what the prompt sampler, the memory-pool read and ProCA cost beside
    the hand-written operators -- is part of the contract (584 in round 5's tree, 478 with ProCA fused and the frequency vectors
    cached, 133 with the prompt sampler as kernels -- csrc/prompt_sampler.hip -- and the self-attention over 550 tokens on the fused
    attention core: tools/launch_sources.py lists them by source line).  The bounds keep them from creeping back: ATen operators
    (dispatch count) and kernels / copies the GPU executes for the whole clip, backbone included (the profiler's device events: 925 in
    round 5, ~800 at the start of round 6, 439 now: profiles/r06_prompted_clip_breakdown_v2.txt)."""
    from torch.utils._python_dispatch import TorchDispatchMode
    from univs_amd import workloads
    swin, head = workloads.build_model(cuda)
    case = dict(workloads.CFG2, H=736, W=1280)
    x = workloads.preprocess(workloads.cfg2_frames()).to(cuda)
    tv0 = workloads.targets_with_entities(case, first_frame_idx=1, n_ent=10)[0]
    tvd = {k: (v.to(cuda) if isinstance(v, torch.Tensor) else v) for k, v in tv0.items()}
    views = ("view", "reshape", "expand", "permute", "transpose", "t.default", "slice", "select", "unsqueeze", "squeeze", "detach", "alias",
             "as_strided", "unbind", "split", "chunk", "_unsafe_view", "empty", "sym_", "size", "stride", "is_", "unflatten", "narrow",
             "movedim", "flatten", "lift_fresh", "_local_scalar_dense", "item", "resize_", "set_", "zeros.default", "result_type", "unfold")

    class Count(TorchDispatchMode):
        n = 0

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func).replace("aten.", "")
            if not any(name.startswith(p) or ("." + p) in name for p in views):
                Count.n += 1
            return func(*args, **(kwargs or {}))
    enc = head.predictor.visual_prompt_sampler.visual_prompt_encoder
    enc.sampler_rng = "auto"                                      # the shipped default (tests/conftest.py pins "reference")
    with torch.no_grad():
        feats = swin(x)
        for _ in range(2):
            head(feats, targets=[dict(tvd)])
        with Count():
            out = head(feats, targets=[dict(tvd)])
    assert out["pred_masks"].shape[1] == 110
    print(f"prompted clip: {Count.n} ATen operators that launch")
    assert Count.n <= 170, Count.n
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    with torch.no_grad():
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            tgl = [dict(tvd)]
            head.prefetch_prompts(tgl, x.shape[0])
            head(swin(x), targets=tgl)
            torch.cuda.synchronize()
    dev_events = [e for e in prof.events() if e.device_type == DeviceType.CUDA]
    print(f"prompted clip: {len(dev_events)} device events (kernels and copies), backbone included")
    assert 250 <= len(dev_events) <= 480, len(dev_events)


def test_head_t10_q200_matches_reference(cuda, golden_dir):
    """The decoder at BASELINE config 5's length (T = 10, 200 queries: 2 000-token spatio-temporal self-attention, class means
    over 10 frames) against the reference head run on reduced-resolution synthetic features (golden g6c): mask logits within
    1e-3, sign-identical; class logits, embeddings, per-query magnitudes and positive counts."""
    g = _g(golden_dir, "g6c_head_t10_q200")
    case = cases.HEAD_CASE_T10
    head = helpers.build_head(case, cuda, return_aux=False)
    with torch.no_grad():
        out = head(_to(cases.backbone_features(case), cuda), targets=_targets_to(cases.targets_first_clip(case), cuda))
    pm = out["pred_masks"].cpu()
    ref = torch.from_numpy(g["pred_masks_q4"])
    err = (pm[:, ::4] - ref).abs().max().item()
    flips = (((pm[:, ::4] > 0) != (ref > 0)) & (ref.abs() > 1e-3)).sum().item()
    e_log = (out["pred_logits"].cpu()[:, :, ::16] - torch.from_numpy(g["pred_logits_k16"])).abs().max().item()
    e_max = (out["pred_logits"].cpu().amax(-1) - torch.from_numpy(g["pred_logits_max"])).abs().max().item()
    e_emb = (out["pred_embds"].cpu()[:, ::4] - torch.from_numpy(g["pred_embds_q4"])).abs().max().item()
    e_abs = (pm.abs().amax(dim=(0, 2, 3, 4)) - torch.from_numpy(g["pred_masks_absmax"])).abs().max().item()
    d_pos = ((pm > 0).sum(dim=(0, 2, 3, 4)) - torch.from_numpy(g["pred_masks_positive"])).abs().max().item()
    print(f"T=10 Q=200 head: mask logits {err:.2e} ({flips} flips), class logits {e_log:.2e} / max {e_max:.2e}, embeddings {e_emb:.2e}, "
          f"per-query |logit| max {e_abs:.2e}, positive-pixel count difference {d_pos}")
    assert err < 1e-3 and flips == 0
    # class logits (not part of the north-star bound): free-running, 200 queries x 9 layers x 10 frames of thresholded attention
    # masks -- an entry whose reference logit sits within rounding of the threshold may land on the other side and moves that
    # ONE query's states by ~1e-3 (the discontinuity test_config4 names entry by entry).  Every query but at most two within
    # 1e-3, those two within 5e-3.
    per_q = (out["pred_logits"].cpu()[:, :, ::16] - torch.from_numpy(g["pred_logits_k16"])).abs().amax(-1)[0]
    off = (per_q >= 1e-3).nonzero().flatten().tolist()
    print(f"queries with a class-logit error >= 1e-3: {off} ({[round(per_q[i].item(), 5) for i in off]})")
    assert len(off) <= 2 and e_log < 5e-3 and e_max < 5e-3 and e_emb < 1e-3 and e_abs < 1e-3
    assert d_pos <= 2          # pixels whose reference logit is within 1e-3 of zero may land on either side


def test_prompt_prefetch_on_a_side_stream_equals_the_inline_sampler(cuda, monkeypatch):
    """The head starts the annotation-only part of the visual-prompt sampler on a side stream before the pixel decoder
    (VisualPromptSampler.prefetch: candidate pixels, feature-resolution masks, the sizes of the reference's randperm
    draws in ONE host round trip); the same clip with that work done inline must give identical outputs and pool."""
    head = helpers.build_head(cases.HEAD_CASE, cuda)
    feats = _to(cases.backbone_features(), cuda)
    sampler = head.predictor.visual_prompt_sampler
    calls = []
    orig = sampler.prefetch
    monkeypatch.setattr(sampler, "prefetch", lambda tv, nf: (calls.append(nf), orig(tv, nf))[1])
    res = []
    for use in (True, False, False):
        targets = _targets_to(cases.targets_with_entities(), cuda)
        if not use:
            monkeypatch.setattr(head.predictor, "prefetch_prompts", lambda *a: None)
        with torch.no_grad():
            torch.manual_seed(3)
            out = head(feats, targets=targets)
        assert "_prompt_prefetch" not in targets[0]
        res.append((out, targets[0]))
    assert calls == [cases.HEAD_CASE["T"]], "the prefetch must have run exactly once, for the first of the three calls"
    # the sampler's own products are deterministic: identical pools; the decoder behind them is compared at the run-to-run
    # spread of two inline runs (library GEMMs with split reductions are not bitwise repeatable)
    for k in ("prompt_feats", "prompt_pe", "prompt_attn_masks"):
        assert torch.equal(res[0][1][k], res[1][1][k]), k
    for k in ("pred_masks", "pred_logits", "pred_embds"):
        d_pf = (res[0][0][k] - res[1][0][k]).abs().max().item()
        d_rr = (res[1][0][k] - res[2][0][k]).abs().max().item()
        print(f"{k}: prefetch vs inline {d_pf:.2e}, inline vs inline {d_rr:.2e}")
        assert d_pf <= max(2 * d_rr, 1e-5), (k, d_pf, d_rr)


def test_g4_g5_prediction_heads_and_teacher_forced_layer(cuda, golden_dir):
    """SURVEY.md Appendix B G4 / G5 through the HIP operators: prediction heads (mask decode + fused attention mask) and
    one teacher-forced decoder layer against tensors captured inside the reference decoder."""
    from univs_amd import ops
    g = _g(golden_dir, "g4_g5_teacher_forced")
    head = helpers.build_head(cases.HEAD_CASE, cuda)
    with torch.no_grad():
        helpers.check_g4_g5(head, g, cuda, ops)


def test_g2_msdeformattn_layer_matches_reference(cuda, golden_dir):
    """SURVEY.md Appendix B G2 through the HIP operators: ONE MSDeformAttn.forward (value / offset / weight projections,
    fused input preparation + LDS-tiled sampling kernel, output projection) and ONE encoder layer in isolation, against
    tensors captured inside the reference pixel decoder."""
    g = _g(golden_dir, "g2_msdeformattn_layer")
    pd = helpers.build_pixel_decoder(cases.HEAD_CASE["shapes"], cuda)
    with torch.no_grad():
        e1, e2 = helpers.check_g2(pd, g, cuda, tol=1e-4)
    assert ops.msda_last_impl() == 2, "the LDS-tiled MSDA kernel must be the one that ran"
    print(f"G2: MSDeformAttn.forward max-abs-err {e1:.2e}, encoder layer {e2:.2e}")


def test_b1_under_the_reference_autocast_context(cuda, golden_dir):
    """Boundary B1 in the reference's real calling context: `train_net.py:334` evaluates under `with autocast():`, so
    `model.backbone(x)` and `model.sem_seg_head(features, targets=...)` (inference_video_entity.py:312,316) run inside an
    fp16 autocast region and may be handed fp16 tensors.  The modules pin themselves to fp32 at their entries
    (layers.fp32_region; the reference does the same for its pixel decoder only, msdeformattn.py:316): Swin-T (g9) and the
    head (g6) reproduce the reference's fp32 CPU outputs at the usual bounds, also when the caller passes fp16 features."""
    swin = helpers.build_swin(cuda)
    head = helpers.build_head(cases.HEAD_CASE, cuda)
    g9, g6 = _g(golden_dir, "g9_swin"), _g(golden_dir, "g6_head_first_clip")
    with torch.no_grad(), torch.autocast("cuda"):
        assert torch.is_autocast_enabled("cuda")
        feats = swin(cases.swin_input().to(cuda))
        out = head(_to(cases.backbone_features(), cuda), targets=_targets_to(cases.targets_first_clip(), cuda))
        # an autocast caller's own ops produce fp16: conv output of the frames -> backbone must up-cast, not raise
        half_in = cases.swin_input().to(cuda).half()
        feats_h = swin(half_in)
    for k in ("res2", "res3", "res4", "res5"):
        assert feats[k].dtype == torch.float32 and feats_h[k].dtype == torch.float32
        assert np.abs(feats[k].cpu().numpy() - g9[k]).max() < 5e-4, k
    helpers.check_head_outputs(out, g6, "", tol=1e-3)
    assert out["pred_masks"].dtype == torch.float32


@pytest.mark.parametrize("autocast", [False, True], ids=["plain", "under_autocast"])
def test_config2_full_size_against_reference(cuda, golden_dir, autocast):
    """BASELINE config 2 (Swin-T, T=5 @ 720p -> 736x1280, 100 queries, first clip): every stage against
    strided samples / checksums of the reference's own CPU run (g12) -- called plainly and inside the reference's
    `with autocast():` evaluation context (train_net.py:334)."""
    g = _g(golden_dir, "g12_cfg2_full_size")
    case = cases.CFG2
    swin = helpers.build_swin(cuda)
    head = helpers.build_head(case, cuda, return_aux=False)
    x = cases.preprocess(cases.cfg2_frames()).to(cuda)
    layers.reset_library_linear_counts()
    with torch.no_grad(), torch.autocast("cuda", enabled=autocast):
        feats = swin(x)
        out = head(feats, targets=_targets_to(cases.targets_first_clip(case), cuda))
    _check_library_linears("cfg2", LIBRARY_LINEARS_ALLOWED["cfg2"])
    for k, v in feats.items():
        err = np.abs(v[:, ::8, ::4, ::4].cpu().numpy() - g["feat_" + k + "_s"]).max()
        assert err < 2e-3, (k, err)
    pm = out["pred_masks"]
    ref_s = g["pred_masks_s"]
    got_s = pm[0, :, :, ::16, ::16].cpu().numpy()
    err = np.abs(got_s - ref_s).max()
    flips = ((got_s > 0) != (ref_s > 0)) & (np.abs(ref_s) > 1e-3)
    print(f"cfg2 pred_masks: max-abs-err {err:.3e}, |ref| max {np.abs(ref_s).max():.2f}, sign flips {flips.sum()}")
    assert err < 1e-3, err      # the north star's bound, absolute
    assert flips.sum() == 0
    # whole-tensor checks: sign map hash (argmax/>0 identical), positive count, mean |logit|
    pmc = pm.cpu()
    sha = hashlib.sha256(np.packbits((pmc > 0).numpy()).tobytes()).digest()
    same_sign_map = bytes(g["pred_masks_sign_sha256"].tobytes()) == sha
    pos = int((pmc > 0).sum())
    print(f"cfg2 sign-map identical: {same_sign_map}; positives {pos} vs {int(g['pred_masks_pos_count'])}; "
          f"|logit|<1e-3 in reference: {int(g['pred_masks_near_zero_1e-3'])}")
    assert abs(pos - int(g["pred_masks_pos_count"])) <= int(g["pred_masks_near_zero_1e-3"])
    assert abs(float(pmc.double().abs().mean()) - float(g["pred_masks_abs_mean"])) < 1e-4
    assert np.abs(out["pred_logits"].cpu().numpy() - g["pred_logits"]).max() < 2e-3
    assert np.abs(out["pred_embds"].cpu().numpy() - g["pred_embds"]).max() < 2e-3


def test_head_gpu_equals_cpu_oracle_path(cuda):
    """Same seeded inputs through the HIP path and through the CPU oracle path (oracle/cpu_path.py)."""
    case = dict(cases.HEAD_CASE, name="head_rt", T=3, H=96, W=160)
    feats = cases.backbone_features(case)
    head_cpu = helpers.build_head(case, "cpu", return_aux=False)
    with cpu_ops(), torch.no_grad():
        ref = head_cpu(feats, targets=cases.targets_first_clip(case))
    head_gpu = helpers.build_head(case, cuda, return_aux=False)
    with torch.no_grad():
        out = head_gpu(_to(feats, cuda), targets=_targets_to(cases.targets_first_clip(case), cuda))
    for k in ("pred_masks", "pred_logits", "pred_embds"):
        err = (out[k].cpu() - ref[k]).abs().max().item()
        assert err < 1e-3, (k, err)
    assert ((out["pred_masks"].cpu() > 0) != (ref["pred_masks"] > 0))[ref["pred_masks"].abs() > 1e-3].sum() == 0


def test_swin_b_window12_matches_reference(cuda, golden_dir):
    """Swin-B (BASELINE config 4's backbone): 144-token windows -> window_attn_f32<9>."""
    g = _g(golden_dir, "g9b_swin_b")
    swin = helpers.build_swin(cuda, variant=cases.SWIN_B)
    with torch.no_grad():
        out = swin(cases.swin_input(cases.SWINB_CASE).to(cuda))
    for k in ("res2", "res3", "res4", "res5"):
        err = np.abs(out[k][:, ::2].cpu().numpy() - g[k]).max()
        assert err < 1e-3, (k, err)


def test_config1_resnet50_plumbing_gpu(cuda):
    """BASELINE config 1 on the device: config-built model (ResNet-50 + head), T=2 @ 256x448, 20 queries."""
    from univs_amd import synth
    from univs_amd.config import get_cfg
    from univs_amd.modeling.build import UniVSHotPath
    cfg = get_cfg()
    cfg.MODEL.BACKBONE.NAME = "build_resnet_backbone"
    cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES = 20
    cfg.INPUT.SAMPLING_FRAME_NUM = 2
    cfg.MODEL.UniVS.CLIP_CLASS_EMBED_PATH = cases.clip_table()
    model = UniVSHotPath(cfg).eval()
    synth.load_synthetic(model)
    model = model.to(cuda)
    frames = synth.synthetic_frames(2, 250, 440, "cfg1/frames").to(cuda)
    out = model(frames, _targets_to(cases.targets_first_clip(dict(cases.HEAD_CASE, T=2)), cuda))
    assert tuple(out["pred_masks"].shape) == (1, 20, 2, 64, 112) and torch.isfinite(out["pred_masks"]).all()
    # same clip through the CPU oracle path
    model_c = UniVSHotPath(cfg).eval()
    synth.load_synthetic(model_c)
    with cpu_ops():
        ref = model_c(frames.cpu(), cases.targets_first_clip(dict(cases.HEAD_CASE, T=2)))
    err = (out["pred_masks"].cpu() - ref["pred_masks"]).abs().max().item()
    assert err < 1e-3, err


def test_clip_loop_on_device_matches_reference(cuda, golden_dir):
    """The whole clip loop on the GPU (HIP operators, device-resident bookkeeping) against the REFERENCE's
    per-clip `targets[0]` states on the 7-frame synthetic video (g11a)."""
    import types

    from tests.test_clip_loop_cpu import compare_states, run_loop
    g = _g(golden_dir, "g11a_clip_loop_model")
    case = cases.LOOP_CASE
    model = types.SimpleNamespace(backbone=helpers.build_swin(cuda), sem_seg_head=helpers.build_head(case, cuda))
    got, results = run_loop(case, model, device=cuda, stability_score_thresh=0.0)
    assert ops.msda_last_impl() in (1, 2)
    compare_states(got, g, tol=1e-3, mask_margin=1e-3)
    assert len(results) == 1 and results[0][0]["masks"].shape[-2:] == case["image_size"]


def test_device_sampler_on_the_gpu(cuda, golden_dir):
    """`sampler="device"` (UNIVS_SAMPLER=device; prompt_encoder.py:290, :369, :506-544) ON the device: the prompt sampler draws with the
    device generator and never calls the host -- no `torch.randperm`, no `.tolist()` inside the head -- over the three clips of the
    7-frame video (memory pool, ProCA).  Its draws differ from the reference's random stream, so what is asserted is what does NOT
    depend on individual draws: the clip schedule, the integer book-keeping of the first clip (no prompts yet: identical to the
    reference golden g11a), the entities found, class logits and mask areas of later clips close to the reference-mode run; sampled
    pixels lie inside their entity's mask (operator level: tests/test_sampler_device_cpu.py); seeded runs repeat bit for bit."""
    import types

    from tests.test_clip_loop_cpu import run_loop
    g = _g(golden_dir, "g11a_clip_loop_model")
    case = cases.LOOP_CASE
    model = types.SimpleNamespace(backbone=helpers.build_swin(cuda), sem_seg_head=helpers.build_head(case, cuda))
    enc = model.sem_seg_head.predictor.visual_prompt_sampler.visual_prompt_encoder
    ref_run, _ = run_loop(case, model, device=cuda, stability_score_thresh=0.0)
    assert enc.sampler_rng == "reference"
    enc.sampler_rng = "device"
    orig_rp, orig_tl = torch.randperm, torch.Tensor.tolist
    head_fwd = model.sem_seg_head.predictor.forward
    inside = {"n": 0}

    def guarded_fwd(*a, **k):
        inside["n"] += 1
        try:
            return head_fwd(*a, **k)
        finally:
            inside["n"] -= 1

    def no_randperm(*a, **k):
        raise AssertionError("torch.randperm in device sampler mode")

    def guarded_tolist(self):
        assert not (inside["n"] and self.is_cuda), "host round trip (.tolist of a device tensor) inside the head in device sampler mode"
        return orig_tl(self)
    model.sem_seg_head.predictor.forward = guarded_fwd
    torch.randperm, torch.Tensor.tolist = no_randperm, guarded_tolist
    try:
        got, results = run_loop(case, model, device=cuda, stability_score_thresh=0.0)
        # (no reset of the device generator: the loop reseeds it at the start of every video from the default generator, which
        # run_loop seeds -- VisualPromptEncoder.begin_video)
        again, _ = run_loop(case, model, device=cuda, stability_score_thresh=0.0)
        enc.sampler_rng = "auto"                                # the SHIPPED default (tests/conftest.py pins "reference"): device draws here
        shipped, _ = run_loop(case, model, device=cuda, stability_score_thresh=0.0)
    finally:
        torch.randperm, torch.Tensor.tolist = orig_rp, orig_tl
        model.sem_seg_head.predictor.forward = head_fwd
        enc.sampler_rng = "reference"
    assert got["clip_first_frames"].tolist() == g["clip_first_frames"].tolist() == [0, 2, 4]
    assert sorted(got) == sorted(ref_run)
    for other in (again, shipped):
        # Seeded: the same video twice gives the same entities and the same pool layout.  (Bit identity of the DRAWS is
        # test_device_sampler_draws_are_reproducible_on_the_gpu; the states of this 64 x 96 video cannot be compared bit for bit:
        # its small Linears and convolutions run on the library, whose GEMMs differ by 1e-5 from run to run -- in REFERENCE sampler
        # mode too, tools/debug_loop_determinism.py -- and a near-threshold decision of the book-keeping may then fall either way.)
        assert sorted(other) == sorted(got)
        for k in got:
            assert got[k].shape == other[k].shape, k
            if k.startswith("clip0_in_") or k.startswith("clip1_in_"):
                if got[k].dtype.is_floating_point and got[k].numel():
                    assert (got[k] - other[k]).abs().max().item() <= 1e-4 * max(1.0, got[k].abs().max().item()), k
    for k in got:
        assert got[k].shape == ref_run[k].shape, k              # same entities, same pool layout
        if k.startswith("clip0_in_") or k.startswith("clip1_in_"):
            # up to the entry of the second clip no prompt token has been sampled: the device mode IS the reference mode there
            if got[k].dtype.is_floating_point:
                assert (got[k] - ref_run[k]).abs().max().item() <= 1e-5 * max(1.0, ref_run[k].abs().max().item()), k
            else:
                assert torch.equal(got[k], ref_run[k]), k
    for k in ("final_ids", "final_first_appear_frame_idxs", "final_frame_indices", "final_occurrence"):
        assert torch.equal(got[k], ref_run[k]), k
    # later clips see different prompt tokens: class logits and mask areas stay close to the reference-mode run
    assert (got["final_logits"] - ref_run["final_logits"]).abs().max().item() < 5e-2
    a, b = (got["final_masks"] > 0.5).float().sum((-2, -1)), (ref_run["final_masks"] > 0.5).float().sum((-2, -1))
    assert ((a - b).abs() <= 0.1 * b.clamp(min=50.0)).all(), (a, b)
    assert torch.isfinite(got["final_mask_logits"]).all() and len(results) == 1


def _cfg4_run(cuda, attn_hook=None):
    """Config 4 at full size through the HIP path; `attn_hook(call_index, our_mask) -> mask to use` wraps the fused
    attention-mask operator (ops.mask_decode_attn: one call per prediction head, 10 per clip)."""
    case = cases.CFG4
    swin = helpers.build_swin(cuda, variant=cases.SWIN_B)
    head = helpers.build_head(case, cuda, return_aux=False, **cases.CFG4_DECODER)
    x = cases.preprocess(cases.cfg2_frames()).to(cuda)
    orig = ops.mask_decode_attn
    calls = {"n": 0}

    def hooked(e, f, deferred=False):
        m = orig(e, f)                  # (the eager form: the masks are compared with / replaced by the reference's tensors)
        k = calls["n"]
        calls["n"] += 1
        return m if attn_hook is None else attn_hook(k, m)
    ops.mask_decode_attn = hooked
    try:
        with torch.no_grad():
            feats = swin(x)
            out = head(feats, targets=_targets_to(cases.cfg4_targets(case), cuda))
    finally:
        ops.mask_decode_attn = orig
    assert calls["n"] == 10
    return feats, out


def _cfg4_ref_mask(gm, i):
    shape = tuple(int(v) for v in gm[f"layer{i}_attn_mask_shape"])
    return np.unpackbits(gm[f"layer{i}_attn_mask_bits"], axis=-1, count=shape[-1]).astype(bool).reshape(shape)


def test_config4_teacher_forced_attention_masks(cuda, golden_dir):
    """BASELINE config 4 with the reference's OWN per-layer bool attention masks supplied (g14b: `memory_mask` of every
    decoder layer's cross-attention, captured inside the unmodified reference, ...decoder_univs.py:390,400-405,555-566).
    With the mask threshold's discontinuity taken out, the north star's bound holds with no escape clause:
    ALL 204 queries within 1e-3 max-abs of the reference's mask logits, no sign flips."""
    gm = _g(golden_dir, "g14b_cfg4_attn_masks")
    case = cases.CFG4

    def force(k, ours):
        if k >= 9:                      # the last head's attention mask is not consumed
            return ours
        ref = torch.from_numpy(_cfg4_ref_mask(gm, k)).to(ours.device)
        assert ref.shape == ours.shape
        return ref
    feats, out = _cfg4_run(cuda, force)
    pm = out["pred_masks"]
    assert pm.shape[1] == case["Q"] + case["n_exp"]
    ref_s = gm["pred_masks_s"]
    got_s = pm[0, :, :, ::16, ::16].cpu().numpy()
    per_q = np.abs(got_s - ref_s).reshape(got_s.shape[0], -1).max(1)
    flips = ((got_s > 0) != (ref_s > 0)) & (np.abs(ref_s) > 1e-3)
    print(f"cfg4 teacher-forced: max-abs-err {per_q.max():.3e} (query {int(per_q.argmax())}), queries over 5e-4: "
          f"{int((per_q > 5e-4).sum())} of {len(per_q)}, sign flips {int(flips.sum())}")
    assert per_q.max() < 1e-3, (int(per_q.argmax()), float(per_q.max()))
    assert flips.sum() == 0


def test_config4_full_size_against_reference(cuda, golden_dir):
    """BASELINE config 4 (Swin-B window 12, grounding with 4 expressions 'sep-blocked', 200 queries, T=5 @ 720p), FREE
    RUNNING: every stage against strided samples / checksums of the reference's own CPU run (g14).

    Every decoder layer thresholds the resized mask logits into a bool attention mask (`sigmoid < 0.5`,
    ...decoder_univs.py:563): an entry whose reference logit is within rounding of 0 can land on the other side in any
    other fp32 implementation, and the query's following cross-attentions then see one key more or less.  The test does
    not grant that as a blanket tolerance; it NAMES the entries from the golden (g14b holds the reference's masks of
    every layer and all resized logits within 1e-2 of the threshold):
      * every entry where our mask differs from the reference's must be one whose reference logit is near the threshold
        (|logit| < 5e-3; the first flip of a query is at rounding level, later ones follow from the first);
      * a query with NO differing entry in any layer must meet the north star's 1e-3; only queries that own a named flipped
        entry may exceed it, and stay below 2e-3;
      * no sign flips in the output anywhere."""
    path = os.path.join(golden_dir, "g14_cfg4_full_size.npz")
    if not os.path.exists(path):
        pytest.skip("g14 golden not generated")
    g = np.load(path)
    gm = _g(golden_dir, "g14b_cfg4_attn_masks")
    case = cases.CFG4
    ours = {}

    def record(k, m):
        ours[k] = m.cpu().numpy()
        return m
    layers.reset_library_linear_counts()
    feats, out = _cfg4_run(cuda, record)
    _check_library_linears("cfg4", LIBRARY_LINEARS_ALLOWED["cfg4"])
    for k, v in feats.items():
        err = np.abs(v[:, ::16, ::4, ::4].cpu().numpy() - g["feat_" + k + "_s"]).max()
        assert err < 3e-3, (k, err)
    # ---- attention masks, layer by layer, against the reference's
    flipped_queries = {}
    for i in range(9):
        ref = _cfg4_ref_mask(gm, i)
        T, Q, HW = ref.shape
        diff = np.flatnonzero(ours[i] != ref)
        if diff.size == 0:
            continue
        near = dict(zip(gm[f"call{i}_near_idx"].tolist(), gm[f"call{i}_near_val"].tolist()))
        for flat in diff.tolist():
            t, q, p = flat // (Q * HW), (flat // HW) % Q, flat % HW
            # a row the reference reset as fully masked (:390) differs wholesale when one of its entries flips
            if flat not in near:
                row_ref, row_our = ref[t, q], ours[i][t, q]
                base = (t * Q + q) * HW
                row_near = [near[j] for j in range(base, base + HW) if j in near]
                assert ((not row_ref.any()) or (not row_our.any())) and row_near, \
                    f"layer {i}: mask entry (t={t}, q={q}, pixel={p}) differs and the reference logit is not near 0"
                if not any(e[0] == i and e[1] == t for e in flipped_queries.get(q, [])):
                    flipped_queries.setdefault(q, []).append((i, t, -1, min(row_near, key=abs)))   # whole-row reset (:390)
                continue
            assert abs(near[flat]) < 5e-3, (i, t, q, p, near[flat])
            flipped_queries.setdefault(q, []).append((i, t, p, near[flat]))
    for q, ent in sorted(flipped_queries.items()):
        i, t, p, v = ent[0]
        print(f"cfg4 flipped attention-mask entry: query {q}, first in layer {i} (frame {t}, pixel {p}, reference logit {v:+.2e}); "
              f"{len(ent)} entries over all layers")
    pm = out["pred_masks"]
    assert pm.shape[1] == case["Q"] + case["n_exp"]
    ref_s = g["pred_masks_s"]
    got_s = pm[0, :, :, ::16, ::16].cpu().numpy()
    err = np.abs(got_s - ref_s).max()
    flips = ((got_s > 0) != (ref_s > 0)) & (np.abs(ref_s) > 1e-3)
    per_q = np.abs(got_s - ref_s).reshape(got_s.shape[0], -1).max(1)
    print(f"cfg4 pred_masks: max-abs-err {err:.3e} (query {int(per_q.argmax())}), |ref| max {np.abs(ref_s).max():.2f}, sign flips "
          f"{flips.sum()}, queries over 5e-4: {int((per_q > 5e-4).sum())} of {len(per_q)}, queries with a flipped mask entry: "
          f"{sorted(flipped_queries)}")
    # (the count is part of the contract: more queries with an entry on the other side of the threshold than the split arithmetic
    # has ever produced means the arithmetic got worse, whatever the clause below allows each of them -- VERDICT r05)
    assert len(flipped_queries) <= CFG4_MAX_FLIPPED_QUERIES, sorted(flipped_queries)
    for q in np.flatnonzero(per_q > 1e-3).tolist():
        assert q in flipped_queries, f"query {q}: {per_q[q]:.2e} > 1e-3 without any attention-mask entry on the other side"
        assert abs(flipped_queries[q][0][3]) < 1e-3, (q, flipped_queries[q][0])
    assert per_q.max() < 2e-3, float(per_q.max())
    assert flips.sum() == 0
    pos = int((pm > 0).sum())
    assert abs(pos - int(g["pred_masks_pos_count"])) <= int(g["pred_masks_near_zero_1e-3"]) + 8
    assert abs(float(pm.double().abs().mean()) - float(g["pred_masks_abs_mean"])) < 1e-4
    assert np.abs(out["pred_logits"].cpu().numpy() - g["pred_logits"]).max() < 3e-3
    assert np.abs(out["pred_embds"][:, :, :, ::4].cpu().numpy() - g["pred_embds"]).max() < 3e-3


def test_config5_swinl_1080p_against_reference(cuda, golden_dir):
    """BASELINE config 5's network (Swin-L window 12 -> window_attn_f32<9>, 200 queries) at 1080p (1088x1920 padded):
    the first two frames against the reference's own CPU run (g19), absolute 1e-3 on the mask logits."""
    g = _g(golden_dir, "g19_cfg5_swinl_1080p")
    case = dict(cases.CFG5, T=cases.CFG5_GOLDEN_T)
    swin = helpers.build_swin(cuda, variant=cases.SWIN_L)
    head = helpers.build_head(case, cuda, return_aux=False)
    x = cases.preprocess(cases.cfg5_frames(case["T"])).to(cuda)
    assert tuple(x.shape[-2:]) == (1088, 1920)
    layers.reset_library_linear_counts()
    with torch.no_grad():
        feats = swin(x)
        out = head(feats, targets=_targets_to(cases.targets_first_clip(case), cuda))
    _check_library_linears("cfg5", LIBRARY_LINEARS_ALLOWED["cfg5"])
    for k, v in feats.items():
        err = np.abs(v[:, ::16, ::4, ::4].cpu().numpy() - g["feat_" + k + "_s"]).max()
        assert err < 3e-3, (k, err)
    pm = out["pred_masks"]
    assert tuple(pm.shape) == (1, case["Q"], case["T"], 272, 480)
    ref_s = g["pred_masks_s"]
    got_s = pm[0, :, :, ::16, ::16].cpu().numpy()
    err = np.abs(got_s - ref_s).max()
    flips = ((got_s > 0) != (ref_s > 0)) & (np.abs(ref_s) > 1e-3)
    print(f"cfg5 (T=2) pred_masks: max-abs-err {err:.3e}, |ref| max {np.abs(ref_s).max():.2f}, sign flips {flips.sum()}")
    assert err < 1e-3, err
    assert flips.sum() == 0
    pos = int((pm > 0).sum())
    assert abs(pos - int(g["pred_masks_pos_count"])) <= int(g["pred_masks_near_zero_1e-3"]) + 8
    assert abs(float(pm.double().abs().mean()) - float(g["pred_masks_abs_mean"])) < 1e-4
    assert np.abs(out["pred_logits"].cpu().numpy() - g["pred_logits"]).max() < 3e-3
    assert np.abs(out["pred_embds"][:, :, :, ::4].cpu().numpy() - g["pred_embds"]).max() < 3e-3


# Tolerances of BASELINE config 5 with the fp16-operand window attention (SwinTransformer.set_attention_mma("f16")) against
# the reference's fp32 CPU run (g19).  24 Swin-L blocks each add an operand-rounding error of ~2^-11 relative to the
# attention output; measured on the GPU: backbone features 4.2e-4 (6.5e-4 against the fp32 model), mask logits 1.44e-3
# (|logit| reaches 14; the fp32 model: 3.7e-4), class logits 3.5e-5, no mask sign differs where |reference logit| > 5e-3.
# The bounds keep a factor ~3-4 over the measurement.
# Yardstick (g19b, oracle/gen_golden.py: g19b_cfg5_reference_autocast): the REFERENCE evaluated the way train_net.py:334 evaluates
# it -- under autocast, emulated on the CPU with fp16 -- sits 3.6e-2 from its own fp32 run on the mask logits (features 5e-3 to
# 1e-2, class logits 2.3e-3, 4 993 mask signs differ beyond |logit| 5e-3): the fp16-operand variant here is ~20x closer to the
# fp32 reference than the reference's own half-precision evaluation; the test asserts that ordering.
CFG5_F16_FEATURE_ATOL = 2e-3
CFG5_F16_MASK_ATOL = 5e-3
CFG5_F16_LOGIT_ATOL = 1e-3


def test_config5_fp16_window_attention_against_reference(cuda, golden_dir):
    """BASELINE config 5 as it is named ("MFMA window-attn, fp16"): Swin-L with the window-attention products on fp16
    operands (window_attn_img_f16<9>), everything else fp32, first two frames at 1080p against the reference's fp32 run (g19)
    under the variant's own tolerance; the fp32 model on the same input bounds what the variant changed."""
    g = _g(golden_dir, "g19_cfg5_swinl_1080p")
    case = dict(cases.CFG5, T=cases.CFG5_GOLDEN_T)
    swin = helpers.build_swin(cuda, variant=cases.SWIN_L, attn_mma="f16")
    assert all(m.mma == "f16" for m in swin.modules() if hasattr(m, "mma"))
    head = helpers.build_head(case, cuda, return_aux=False)
    x = cases.preprocess(cases.cfg5_frames(case["T"])).to(cuda)
    with torch.no_grad():
        feats = swin(x)
        out = head(feats, targets=_targets_to(cases.targets_first_clip(case), cuda))
        feats32 = swin.set_attention_mma("f32")(x)
    ferr = {k: float(np.abs(v[:, ::16, ::4, ::4].cpu().numpy() - g["feat_" + k + "_s"]).max()) for k, v in feats.items()}
    fdiff = {k: float((feats[k] - feats32[k]).abs().max()) for k in feats}
    ref_s = g["pred_masks_s"]
    got_s = out["pred_masks"][0, :, :, ::16, ::16].cpu().numpy()
    err = np.abs(got_s - ref_s).max()
    flips = ((got_s > 0) != (ref_s > 0)) & (np.abs(ref_s) > CFG5_F16_MASK_ATOL)
    lerr = np.abs(out["pred_logits"].cpu().numpy() - g["pred_logits"]).max()
    print(f"cfg5 fp16 window attention (T=2): features vs reference {ferr}, vs the fp32 model {fdiff}; pred_masks max-abs-err "
          f"{err:.3e} (|ref| max {np.abs(ref_s).max():.2f}), sign flips beyond the tolerance {flips.sum()}, pred_logits {lerr:.3e}")
    assert max(fdiff.values()) > 0, "the fp16 variant must be the kernel that ran"
    assert max(ferr.values()) < CFG5_F16_FEATURE_ATOL, ferr
    assert err < CFG5_F16_MASK_ATOL, err
    assert flips.sum() == 0
    assert lerr < CFG5_F16_LOGIT_ATOL, lerr
    y = _g(golden_dir, "g19b_cfg5_reference_autocast")
    print(f"yardstick: the reference under (emulated) autocast vs its own fp32 run: pred_masks {float(y['pred_masks_err_s']):.3e} on the same "
          f"samples ({float(y['pred_masks_err_full']):.3e} over the full tensor), features "
          f"{max(float(y['feat_' + k + '_err']) for k in feats):.3e}, pred_logits {float(y['pred_logits_err']):.3e}")
    assert err < float(y["pred_masks_err_s"]) and max(ferr.values()) < max(float(y["feat_" + k + "_err"]) for k in feats)


def test_config5_full_clip_properties(cuda):
    """BASELINE config 5 at FULL size on the GPU (Swin-L, T=10 @ 1080p, 200 queries; the reference's CPU run of this clip
    needs > 100 GB): size-independent properties of the hot operators on the tensors the model really produces --
    finite outputs; the head-major MSDeformAttn kernel (msda_heads.hip) covers the 1080p geometry and == the generic
    kernel on the same layer's operands un-packed to the standard layouts; oracle C on every 37th query of the first
    layer; full-resolution mask decode == fp64 einsum on a strided subset; the first two frames' features == the T=2 run
    (frames are independent in the backbone)."""
    from oracle import c_ops
    case = cases.CFG5
    swin = helpers.build_swin(cuda, variant=cases.SWIN_L)
    head = helpers.build_head(case, cuda, return_aux=False)
    x = cases.preprocess(cases.cfg5_frames(case["T"])).to(cuda)
    seen = {"msda": [], "dec": []}
    orig_heads, orig_dec = ops.msda_forward_heads, ops.mask_decode

    def heads_hook(vhm, qhm, ref_q, shapes, lsi, M, P=4):
        out = orig_heads(vhm, qhm, ref_q, shapes, lsi, M, P)
        assert out is not None and ops.msda_last_tiled_generation() == 6, "the head-major kernel must cover the 1080p geometry"
        if not seen["msda"]:
            seen["msda"].append((vhm, qhm, ref_q, shapes, lsi, out))
        return out

    def dec_hook(e, f):
        out = orig_dec(e, f)
        seen["dec"].append((e, f, out, ops.mask_decode_last_impl()))
        return out
    ops.msda_forward_heads, ops.mask_decode = heads_hook, dec_hook
    try:
        with torch.no_grad():
            feats = swin(x)
            out = head(feats, targets=_targets_to(cases.targets_first_clip(case), cuda))
    finally:
        ops.msda_forward_heads, ops.mask_decode = orig_heads, orig_dec
    pm = out["pred_masks"]
    assert tuple(pm.shape) == (1, 200, 10, 272, 480) and torch.isfinite(pm).all()
    assert torch.isfinite(out["pred_logits"]).all() and torch.isfinite(out["pred_embds"]).all()
    # MSDA: S = 34*60 + 68*120 + 136*240 = 42840 tokens per frame.  Un-pack the head-major operands to the standard layouts
    vhm, qhm, ref_q, shapes, lsi, got = seen["msda"][0]
    N, M, S, D = vhm.shape
    L, P = len(shapes), 4
    assert S == 42840 and N == 10 and D == 32
    value = vhm.permute(0, 2, 1, 3).contiguous()
    order = ops.msda_level_order(shapes)
    q = qhm.view(N, M, S, P, 3 * L)
    off = torch.empty((N, S, M, L, P, 2), device=cuda)
    lg = torch.empty((N, S, M, L, P), device=cuda)
    for kk, l in enumerate(order):
        off[:, :, :, l] = q[..., 2 * kk:2 * kk + 2].permute(0, 2, 1, 3, 4)
        lg[:, :, :, l] = q[..., 2 * L + kk].permute(0, 2, 1, 3)
    norm = torch.tensor([[w_, h_] for (h_, w_) in shapes], dtype=torch.float32, device=cuda).view(1, 1, 1, L, 1, 2)
    loc = (ref_q.view(-1, S, 1, 1, 1, 2) + off / norm).contiguous()
    attn = torch.softmax(lg.reshape(N, S, M, L * P), -1).view(N, S, M, L, P).contiguous()
    with ops.configured(msda_impl=1):
        generic = ops.ms_deform_attn_forward(value[:3], shapes, lsi, loc[:3], attn[:3])
    assert (got[:3] - generic).abs().max().item() < 3e-5
    tiled2 = ops.ms_deform_attn_forward(value[:3], shapes, lsi, loc[:3], attn[:3])
    assert ops.msda_last_tiled_generation() == 2 and (tiled2 - generic).abs().max().item() < 3e-5
    sub = torch.arange(0, S, 37, device=cuda)
    ref = c_ops.msda_forward(value[:2].cpu().numpy(), shapes, lsi, loc[:2, sub].contiguous().cpu().numpy(),
                             attn[:2, sub].contiguous().cpu().numpy())
    assert np.abs(got[:2, sub].cpu().numpy() - ref).max() < 3e-5
    # mask decode (the last call is the full-resolution one that feeds pred_masks)
    e, f, dec, impl = seen["dec"][-1]
    assert tuple(f.shape) == (10, 256, 272, 480) and impl == 2
    ref64 = torch.einsum("tqc,tchw->qthw", e.double(), f[:, :, ::17, ::13].double())
    assert (dec[:, :, ::17, ::13].double() - ref64).abs().max().item() < 1e-4
    # frames are independent in the backbone: the first two frames equal the T=2 run
    with torch.no_grad():
        feats2 = swin(x[:2])
    for k in feats:
        assert (feats[k][:2] - feats2[k]).abs().max().item() < 1e-4, k


def test_config3_long_video_teacher_forced_sampler_strict(cuda, golden_dir):
    """BASELINE config 3 on one GPU with the prompt sampler teacher-forced: the 40-frame 720p video through the sliding
    5-frame clip loop (36 clips, memory pool carried along) with the sampler REPLAYING the pixels the reference sampled
    (g20 `draws_*`: recorded inside the reference's get_mask_prompt calls, prompt_encoder.py:420-481, SURVEY section 7 "hard
    parts").  The sizes of the reference's `randperm` draws are pixel counts of thresholded masks, so without this one
    near-threshold pixel changes every token of a clip; with the draws supplied the two implementations stay on ONE
    trajectory and the north star's bound is asserted for ALL 36 clips and the final state, no escape clause:
    integers exact, mask logits / prompt memory / embeddings within 1e-3, areas within the near-threshold pixel count."""
    import types

    from tests.test_clip_loop_cpu import compare_reduced_states, recorded_draws, run_loop
    g = _g(golden_dir, "g20_cfg3_long_video")
    case = cases.CFG3_LOOP
    model = types.SimpleNamespace(backbone=helpers.build_swin(cuda), sem_seg_head=helpers.build_head(case, cuda))
    draws = recorded_draws(g)
    assert len(draws) == len(g["draws_clip"]) and set(g["draws_clip"].tolist()) == set(range(1, 36))
    got, results = run_loop(case, model, device=cuda, replay=draws, stability_score_thresh=0.0, clip_stride=1)
    assert got["clip_first_frames"].tolist() == g["clip_first_frames"].tolist() and len(g["clip_first_frames"]) == 36

    class View(dict):
        @property
        def files(self):
            return list(self.keys())

    per_clip, failures = [], []
    for c in list(range(36)) + ["final"]:
        keys = [k for k in g.files if k.startswith(f"clip{c}_in_" if c != "final" else "final_")]
        ref, val = View({k: g[k] for k in keys}), {k: got[k] for k in keys}
        k = f"clip{c}_in_mask_logits_s"
        if k in ref and ref[k].size:
            per_clip.append(f"{c}:{np.abs(val[k].numpy() - ref[k]).max():.1e}")
        try:
            compare_reduced_states(val, ref, tol=1e-3)
        except AssertionError as e:
            failures.append((c, str(e)[:400]))
    print("cfg3 teacher-forced sampler: max |mask logit| error per clip:", " ".join(per_clip))
    assert not failures, failures[:3]


def test_config3_long_video_on_device_matches_reference(cuda, golden_dir):
    """BASELINE config 3 on one GPU: the 40-frame 720p video through the sliding 5-frame clip loop (36 clips at the
    reference's default stride 1, prompt memory pool carried from clip to clip) against the REFERENCE's loop (g20: reduced
    per-clip states).

    What can be asserted over 36 clips of a feedback loop: the prompt sampler draws `randperm` over the entity's candidate
    pixels, so ONE pixel whose logit rounds to the other side of 0 changes the candidate count and with it every sampled
    token of that clip -- from there on two fp32 implementations follow different (equally valid) trajectories.  So:
      * every integer of the bookkeeping (entity ids, first-appearance frames, frame indices, occurrence counts, class
        argmax, attention-mask fractions) is exact over ALL 36 clips;
      * until the first such event the states agree to the north star's 1e-3 (measured: clips 0-4, max 1.8e-4, areas
        within the near-threshold pixel count) -- at least the first four clips must;
      * afterwards mask areas stay within 1 % and the class logits (which do not depend on individual pixels) within 1e-3."""
    import types

    from tests.test_clip_loop_cpu import compare_reduced_states, run_loop
    g = _g(golden_dir, "g20_cfg3_long_video")
    case = cases.CFG3_LOOP
    model = types.SimpleNamespace(backbone=helpers.build_swin(cuda), sem_seg_head=helpers.build_head(case, cuda))
    got, results = run_loop(case, model, device=cuda, stability_score_thresh=0.0, clip_stride=1)
    assert got["clip_first_frames"].tolist() == g["clip_first_frames"].tolist() and len(g["clip_first_frames"]) == 36
    assert len(results) >= 1      # one list of result records per finished backbone window

    class View(dict):      # a dict with the `.files` of an npz
        @property
        def files(self):
            return list(self.keys())

    def robust(k):         # fields that do not depend on individual near-threshold pixels
        return not any(t in k for t in ("logits_s", "prompt_pe", "prompt_feats", "embds", "quality", "boxes"))

    first_loose = None
    per_clip = []
    for c in list(range(36)) + ["final"]:
        keys = [k for k in g.files if k.startswith(f"clip{c}_in_" if c != "final" else "final_")]
        ref, val = View({k: g[k] for k in keys}), {k: got[k] for k in keys}
        try:
            compare_reduced_states(val, ref, tol=1e-3)
            tight = True
        except AssertionError:
            tight = False
        k = f"clip{c}_in_mask_logits_s"
        if k in ref and ref[k].size:
            per_clip.append(f"{c}:{np.abs(val[k].numpy() - ref[k]).max():.1e}")
        if not tight:
            if first_loose is None:
                first_loose = c
            # the loose regime: integers exact (inside compare), areas 1 %, class-logit maxima 1e-3
            compare_reduced_states({k: v for k, v in val.items() if robust(k)}, View({k: v for k, v in ref.items() if robust(k)}),
                                   tol=1e-3, area_rel=1e-2)
    print("cfg3 long video: first clip outside 1e-3:", first_loose, "| max |mask logit| error per clip:", " ".join(per_clip))
    assert first_loose is None or first_loose == "final" or first_loose >= 4, first_loose


@pytest.mark.parametrize("name,cfg,tol,stride", [("g16b_text_encoder_small", cases.TEXT_SMALL, 1e-4, 1),
                                                 ("g16c_text_encoder_full", cases.TEXT_FULL, 5e-4, 4)],
                         ids=["small", "rn50x4"])
def test_text_encoder_matches_reference(cuda, golden_dir, name, cfg, tol, stride):
    """CLIP text tower + TextPromptEncoder.get_expression_prompt on the device (HIP LayerNorm / masked softmax) against
    the reference's outputs (oracle/gen_golden.py: g16b / g16c)."""
    from tests.test_language_cpu import check_text_encoder
    err = check_text_encoder(_g(golden_dir, name), cfg, cuda, tol, stride)
    print(f"text encoder {name}: max abs err {err:.2e}")


def test_rle_boundaries_on_device_match_host(cuda):
    """results.rle_encode_masks finds the run boundaries on the device: same strings as from host tensors, decode == mask."""
    from univs_amd.inference import results as R
    m = (torch.rand(6, 736, 1280, generator=torch.Generator().manual_seed(0)) > 0.5)
    m[0] = False
    m[1] = True
    m[2, 100:300, 200:900] = True
    m[2, :100] = False
    host = R.rle_encode_masks(m[:3])
    dev = R.rle_encode_masks(m[:3].to(cuda))
    assert host == dev and host[0]["counts"] == R.rle_encode_masks(torch.zeros(1, 736, 1280, dtype=torch.bool))[0]["counts"]
    assert np.array_equal(R.rle_decode(dev[2]), m[2].numpy().astype(np.uint8))
    noisy = R.rle_encode_masks(m[3:].to(cuda))
    assert all(np.array_equal(R.rle_decode(r), m[3 + i].numpy().astype(np.uint8)) for i, r in enumerate(noisy))
