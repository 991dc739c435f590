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
