"""CPU: host logic of the product modules (univs_amd/modeling) against the golden outputs of the REAL
reference (tests/golden/, produced by oracle/gen_golden.py).  The four HIP operators have no CPU
implementation, so these tests substitute the oracle's CPU stand-ins for them (oracle/cpu_path.py) --
everything else (module graph, caching, prompt encoder / memory pool bookkeeping, state-dict layout) is
the product code.  The same checks run against the HIP operators in tests/test_modules_gpu.py."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.cpu_path import cpu_ops
from tests import cases, helpers
from univs_amd import synth


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def test_state_dict_layout_matches_reference(golden_dir):
    layout = json.load(open(os.path.join(golden_dir, "state_dict_layout.json")))["layout"]
    swin = helpers.build_swin()
    head = helpers.build_head(cases.HEAD_CASE, text_to_image=True)
    mine = {"backbone." + k: list(v.shape) for k, v in swin.state_dict().items()}
    mine.update({"sem_seg_head." + k: list(v.shape) for k, v in head.state_dict().items()})
    assert sorted(mine) == sorted(layout)
    for k in layout:
        assert mine[k] == layout[k], k


def test_position_embeddings(golden_dir):
    from univs_amd.modeling.position_encoding import (PositionEmbeddingSine, PositionEmbeddingSine3D,
                                                      PositionEmbeddingSine3DArbitraryT)
    g = _g(golden_dir, "g10_position_embeddings")
    x = torch.zeros(2, 8, 5, 7)
    x5 = torch.zeros(1, 3, 8, 5, 7)
    pts = synth.uniform("pe/pts", (6, 2), 0.0, 1.0)
    arb = PositionEmbeddingSine3DArbitraryT(4, normalize=True)
    fix = PositionEmbeddingSine3D(4, normalize=True)
    got = {
        "sine2d": PositionEmbeddingSine(4, normalize=True)(x),
        "arb3d": arb(x5, torch.tensor([[4, 5, 9]])),
        "arb3d_default_t": arb(x5),
        "arb_points": arb.forward_points_with_size((3, 40, 56), pts, 7),
        "arb_points_vec": arb.forward_points_with_size((3, 40, 56), pts, torch.tensor([2, 3, 4])),
        "fix3d": fix(x5),
        "fix_points": fix.forward_points_with_size((3, 40, 56), pts),
    }
    for k, v in got.items():
        assert tuple(v.shape) == g[k].shape, k
        assert np.abs(v.numpy() - g[k]).max() < 2e-6, k


def test_swin_matches_reference(golden_dir):
    g = _g(golden_dir, "g9_swin")
    swin = helpers.build_swin()
    with cpu_ops(), torch.no_grad():
        out = swin(cases.swin_input())
    for k in ("res2", "res3", "res4", "res5"):
        err = np.abs(out[k].numpy() - g[k]).max()
        assert err < 2e-4, (k, err)


def test_pixel_decoder_matches_reference(golden_dir):
    g = _g(golden_dir, "g3_pixel_decoder")
    pd = helpers.build_pixel_decoder(cases.HEAD_CASE["shapes"])
    with cpu_ops(), torch.no_grad():
        mf, mf_bfe, enc0, ms = pd.forward_features(cases.backbone_features())
    got = dict(mask_features=mf, mask_features_bfe_conv=mf_bfe, enc0=enc0, ms0=ms[0], ms1=ms[1], ms2=ms[2])
    for k, v in got.items():
        err = np.abs(v.numpy() - g[k]).max()
        assert err < 2e-4, (k, err)


def test_head_t10_q200_matches_reference(golden_dir):
    """Host logic at BASELINE config 5's decoder length (T = 10, 200 queries) against golden g6c (the reference head on
    reduced-resolution features): the CPU path of the same module tree (oracle stand-ins for the HIP operators)."""
    g = _g(golden_dir, "g6c_head_t10_q200")
    case = cases.HEAD_CASE_T10
    head = helpers.build_head(case, return_aux=False)
    with cpu_ops(), torch.no_grad():
        out = head(cases.backbone_features(case), targets=cases.targets_first_clip(case))
    pm, ref = out["pred_masks"], torch.from_numpy(g["pred_masks_q4"])
    assert (pm[:, ::4] - ref).abs().max().item() < 1e-3
    assert (((pm[:, ::4] > 0) != (ref > 0)) & (ref.abs() > 1e-3)).sum().item() == 0
    tol_log = 1e-3 * max(1.0, float(np.abs(g["pred_logits_k16"]).max()) / 10.0)
    assert (out["pred_logits"][:, :, ::16] - torch.from_numpy(g["pred_logits_k16"])).abs().max().item() < tol_log
    assert (out["pred_embds"][:, ::4] - torch.from_numpy(g["pred_embds_q4"])).abs().max().item() < 1e-3


@pytest.mark.parametrize("name,dec_over,targets_fn,seed", helpers.HEAD_SCENARIOS, ids=[s[0] for s in helpers.HEAD_SCENARIOS])
def test_head_matches_reference(golden_dir, name, dec_over, targets_fn, seed):
    g = _g(golden_dir, name)
    head = helpers.build_head(cases.HEAD_CASE, **dec_over)
    targets = targets_fn()
    with cpu_ops(), torch.no_grad():
        if seed is not None:
            torch.manual_seed(seed)
        out = head(cases.backbone_features(), targets=targets)
    helpers.check_head_outputs(out, g, "", tol=1e-3)
    if name == "g7_head_visual_prompts":
        for k in ("prompt_feats", "prompt_pe"):
            assert np.abs(targets[0][k].numpy() - g["pool_" + k]).max() < 1e-4, k
        assert (targets[0]["prompt_attn_masks"].numpy() == g["pool_prompt_attn_masks"]).all()
        # third clip on the same targets dict (memory-pool update + read)
        helpers.advance_to_third_clip(targets)
        with cpu_ops(), torch.no_grad():
            torch.manual_seed(1)
            out3 = head(cases.backbone_features(), targets=targets)
        helpers.check_head_outputs(out3, g, "clip3_", tol=1e-3)
        for k in ("prompt_feats", "prompt_pe"):
            assert np.abs(targets[0][k].numpy() - g["clip3_pool_" + k]).max() < 1e-4, k


def test_g4_g5_prediction_heads_and_teacher_forced_layer(golden_dir):
    """the prediction heads (class logits, mask logits, bool attention mask incl. the
    all-True-row reset) and one decoder layer with every input supplied, against tensors captured inside the reference."""
    from oracle import cpu_path
    g = _g(golden_dir, "g4_g5_teacher_forced")
    head = helpers.build_head(cases.HEAD_CASE)
    with cpu_ops(), torch.no_grad():
        helpers.check_g4_g5(head, g, "cpu", cpu_path)


def test_config1_resnet50_plumbing_cpu():
    """BASELINE config 1: ResNet-50 UniVS, 1 clip x T=2 frames @ 256x448, 20 queries, CPU path (plumbing).
    ResNet parity is unpinned (detectron2 source is not in the reference tree): shapes, strides and the
    Detectron2 key layout are checked, then the clip runs end to end through the config-built model."""
    from univs_amd.config import get_cfg
    from univs_amd.modeling.build import UniVSHotPath
    cfg = get_cfg()
    cfg.MODEL.BACKBONE.NAME = "build_resnet_backbone"
    cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES = 20
    cfg.INPUT.SAMPLING_FRAME_NUM = 2
    cfg.MODEL.UniVS.CLIP_CLASS_EMBED_PATH = cases.clip_table()
    model = UniVSHotPath(cfg).eval()
    synth.load_synthetic(model)
    shp = model.backbone.output_shape()
    assert {k: (v.channels, v.stride) for k, v in shp.items()} == \
        {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)}
    keys = set(model.backbone.state_dict())
    for k in ("stem.conv1.weight", "stem.conv1.norm.running_var", "res2.0.shortcut.weight", "res2.0.conv2.norm.bias",
              "res4.5.conv3.weight", "res5.2.conv1.norm.running_mean"):
        assert k in keys, k
    frames = synth.synthetic_frames(2, 250, 440, "cfg1/frames")     # pads to 256x448
    case = dict(cases.HEAD_CASE, T=2)
    with cpu_ops(), torch.no_grad():
        x = model.preprocess(frames)
        assert tuple(x.shape) == (2, 3, 256, 448)
        feats = model.backbone(x)
        assert tuple(feats["res5"].shape) == (2, 2048, 8, 14) and tuple(feats["res2"].shape) == (2, 256, 64, 112)
        out = model.sem_seg_head(feats, targets=cases.targets_first_clip(case))
    assert tuple(out["pred_masks"].shape) == (1, 20, 2, 64, 112)
    assert tuple(out["pred_logits"].shape) == (1, 20, 3938)
    assert torch.isfinite(out["pred_masks"]).all() and out["aux_outputs"] == []


def test_swin_b_window12_matches_reference(golden_dir):
    g = _g(golden_dir, "g9b_swin_b")
    swin = helpers.build_swin(variant=cases.SWIN_B)
    with cpu_ops(), torch.no_grad():
        out = swin(cases.swin_input(cases.SWINB_CASE))
    for k in ("res2", "res3", "res4", "res5"):
        err = np.abs(out[k][:, ::2].numpy() - g[k]).max()
        assert err < 5e-4, (k, err)


def test_g2_msdeformattn_layer_matches_reference(golden_dir):
    """Appendix B G2 on the CPU oracle path: one MSDeformAttn.forward + one encoder layer in isolation."""
    g = _g(golden_dir, "g2_msdeformattn_layer")
    pd = helpers.build_pixel_decoder(cases.HEAD_CASE["shapes"])
    with cpu_ops(), torch.no_grad():
        helpers.check_g2(pd, g, "cpu")


def test_attention_layers_match_torch_nn():
    """layers.MultiheadAttention with the arguments the decoder layers now pass (`query_add`: the position embedding added inside the
    projection; `residual` + `norm`: the post-norm tail inside the output projection; precomputed key / value projections) against
    torch.nn.MultiheadAttention + nn.LayerNorm composed as the reference composes them (transformer_layers.py:30-46, :95-115), and
    layers.MLP(transpose01=True) against its own transposed result -- the host-side wiring of the few-rows kernels, on the CPU path."""
    from univs_amd import layers
    from univs_amd.modeling.transformer_decoder import transformer_layers as tl
    torch.manual_seed(0)
    E, H, L, S, N = 64, 2, 10, 24, 3
    ref = torch.nn.MultiheadAttention(E, H).eval()
    norm = torch.nn.LayerNorm(E).eval()
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.2, 0.2)
    tgt, pos = torch.randn(L, N, E), torch.randn(L, N, E)
    mem, mpos = torch.randn(S, N, E), torch.randn(S, N, E)
    with cpu_ops(), torch.no_grad():
        # self-attention, post-norm: norm(tgt + attn(q = k = tgt + pos, v = tgt))
        sa = tl.SelfAttentionLayer(E, H).eval()
        sa.self_attn.load_state_dict(ref.state_dict())
        sa.norm.load_state_dict(norm.state_dict())
        mask = torch.rand(L, L) < 0.3
        mask[torch.arange(L), torch.arange(L)] = False
        want = norm(tgt + ref(tgt + pos, tgt + pos, tgt, attn_mask=mask)[0])
        assert (sa(tgt, tgt_mask=mask, query_pos=pos) - want).abs().max() < 1e-5
        assert (sa(tgt, tgt_mask=mask) - norm(tgt + ref(tgt, tgt, tgt, attn_mask=mask)[0])).abs().max() < 1e-5
        # cross-attention, post-norm, per-frame masks shared by the heads; key = memory + pos given; then with precomputed K / V projections
        ca = tl.CrossAttentionLayer(E, H).eval()
        ca.multihead_attn.load_state_dict(ref.state_dict())
        ca.norm.load_state_dict(norm.state_dict())
        cm = torch.rand(N, L, S) < 0.4
        cm[:, :, 0] = False
        want = norm(tgt + ref(tgt + pos, mem + mpos, mem, attn_mask=cm.repeat_interleave(H, 0))[0])
        got = ca(tgt, mem, memory_mask=cm, pos=mpos, query_pos=pos)
        assert (got - want).abs().max() < 1e-5
        w, b = ref.in_proj_weight, ref.in_proj_bias
        kv = (torch.nn.functional.linear(mem + mpos, w[E:2 * E], b[E:2 * E]), torch.nn.functional.linear(mem, w[2 * E:], b[2 * E:]))
        assert (ca(tgt, mem, memory_mask=cm, query_pos=pos, kv=kv) - want).abs().max() < 1e-5
        # the FFN's post-norm tail and the mask-embedding MLP's transposed result
        ffn = tl.FFNLayer(E, 128).eval()
        x = torch.randn(L, N, E)
        want = ffn.norm(x + ffn.linear2(torch.relu(ffn.linear1(x))))
        assert (ffn(x) - want).abs().max() < 1e-5
        mlp = layers.MLP(E, E, 32, 3).eval()
        assert torch.equal(mlp(x, transpose01=True), mlp(x).transpose(0, 1))
