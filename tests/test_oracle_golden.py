"""CPU: the oracle (oracle/ops_ref.c) against the golden vectors produced by the real reference
(tests/golden/*.npz, written by oracle/gen_golden.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import c_ops
from tests import cases
from univs_amd import synth


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def test_g0_msda_kat_matches_reference(golden_dir):
    """Shapes/seed of the reference's own KAT, ops/test.py:24-31; tolerances from :43 and :59."""
    g = _load(golden_dir, "g0_msda_kat")
    for tag, dt in (("double", np.float64), ("float", np.float32)):
        out = c_ops.msda_forward(g[f"value_{tag}"].astype(dt), g["shapes"], g["level_start_index"],
                                 g[f"loc_{tag}"].astype(dt), g[f"attn_{tag}"].astype(dt))
        ref = g[f"out_{tag}"]
        if tag == "double":
            assert np.allclose(out, ref)                       # ops/test.py:43 (torch.allclose default)
            assert np.abs(out - ref).max() < 1e-15
        else:
            assert np.allclose(out, ref, rtol=1e-2, atol=1e-3)  # ops/test.py:59
            assert np.abs(out - ref).max() < 1e-8


@pytest.mark.parametrize("case", cases.MSDA_CASES, ids=lambda c: c["name"])
def test_g1_msda_geometry_matches_reference(golden_dir, case):
    g = _load(golden_dir, "g1_msda_geometry")
    value, shapes, lsi, loc, attn = cases.msda_inputs(case)
    out = c_ops.msda_forward(value.numpy(), shapes, lsi, loc.numpy(), attn.numpy())
    sub = cases.msda_query_subset(case, out.shape[1])
    ref = g[f"{case['name']}/out_subset"]
    err = np.abs(out[:, sub] - ref).max()
    assert err < 2e-5, err   # fp32 vs grid_sample fp32: re-association only
    assert abs(out.astype(np.float64).sum() - float(g[f"{case['name']}/sum"])) < 1e-2
    assert abs(np.abs(out.astype(np.float64)).sum() - float(g[f"{case['name']}/abs_sum"])) < 1e-1


def test_msda_zero_padding_outside_level():
    """A location more than one pixel outside contributes exactly 0 (cuh:293); half a pixel outside
    keeps the in-range corners only (cuh:38-89)."""
    shapes = [(2, 2)]
    value = np.arange(4, dtype=np.float64).reshape(1, 4, 1, 1) + 1.0
    loc = np.array([[-0.6, 0.5], [0.0, 0.0], [1.0, 1.0], [2.0, 0.5]], dtype=np.float64).reshape(1, 1, 1, 1, 4, 2)
    attn = np.eye(4, dtype=np.float64)
    outs = [c_ops.msda_forward(value, shapes, [0], loc, attn[i].reshape(1, 1, 1, 1, 4))[0, 0, 0] for i in range(4)]
    assert outs[0] == 0.0                 # x*W-0.5 = -1.7 <= -1
    assert outs[1] == pytest.approx(0.25 * 1.0)   # corner (0,0) with weight 0.5*0.5
    assert outs[2] == pytest.approx(0.25 * 4.0)   # corner (1,1)
    assert outs[3] == 0.0                 # x*W-0.5 = 3.5 >= W


@pytest.mark.parametrize("case", cases.WINATTN_CASES, ids=lambda c: c["name"])
def test_window_attention_matches_reference(golden_dir, case):
    g = _load(golden_dir, "g_window_attention")
    x, mask = cases.winattn_inputs(case)
    dim, nH, win = case["dim"], case["heads"], case["win"]
    ntok = win * win
    p = case["name"] + "."
    w_qkv = synth.make_param(p + "qkv.weight", (3 * dim, dim)); b_qkv = synth.make_param(p + "qkv.bias", (3 * dim,))
    w_proj = synth.make_param(p + "proj.weight", (dim, dim)); b_proj = synth.make_param(p + "proj.bias", (dim,))
    table = synth.make_param(p + "relative_position_bias_table", ((2 * win - 1) ** 2, nH))
    # relative position index, restated from swin.py:110-120
    ch, cw = torch.meshgrid(torch.arange(win), torch.arange(win), indexing="ij")
    coords = torch.stack([ch.reshape(-1), cw.reshape(-1)])
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0) + (win - 1)
    index = rel[..., 0] * (2 * win - 1) + rel[..., 1]
    bias = table[index.reshape(-1)].view(ntok, ntok, nH).permute(2, 0, 1).contiguous()
    qkv = (x @ w_qkv.t() + b_qkv).view(x.shape[0], ntok, 3, nH, dim // nH)
    core = c_ops.window_attention(qkv.numpy(), bias.numpy(), None if mask is None else mask.numpy(),
                                  (dim // nH) ** -0.5)
    y = torch.from_numpy(core) @ w_proj.t() + b_proj
    err = (y - torch.from_numpy(g[f"{case['name']}/out"])).abs().max().item()
    assert err < 2e-5, err


@pytest.mark.parametrize("case", cases.MASKDEC_CASES, ids=lambda c: c["name"])
def test_mask_decode_matches_einsum(case):
    """The contraction of ...decoder_univs.py:527-528, against torch.einsum in float64."""
    e, f = cases.maskdec_inputs(case)
    out = c_ops.mask_decode(e.numpy(), f.numpy())
    ref = torch.einsum("tqc,tchw->qthw", e.double(), f.double()).numpy()
    assert np.abs(out - ref).max() < 1e-4


def test_attn_mask_rule():
    x = np.array([[[-1.0, 2.0, -0.0, 0.0], [-1.0, -2.0, -3.0, -4.0]]], dtype=np.float32)
    m = c_ops.attn_mask_from_logits(x)
    assert m[0, 0].tolist() == [True, False, False, False]   # sigmoid(0) = 0.5 is not < 0.5
    assert m[0, 1].tolist() == [False] * 4                  # fully masked row is reset (:390)


def test_msda_torch_oracle_backward_matches_reference(golden_dir):
    """oracle/msda_torch.py (the differentiable restatement used by the GPU backward tests) against the
    REAL reference's core + autograd (g13)."""
    from oracle import msda_torch
    g = np.load(os.path.join(golden_dir, "g13_msda_backward.npz"))
    case = cases.MSDA_BWD_CASE
    value, shapes, lsi, loc, attn = cases.msda_inputs(case)
    go = synth.normal("msda_bwd/go/" + case["name"], (value.shape[0], loc.shape[1], value.shape[2] * value.shape[3]))
    out = msda_torch.forward(value.double(), shapes, loc.double(), attn.double())
    assert np.abs(out.numpy() - g["out"]).max() < 1e-5
    for name, t in zip(("grad_value", "grad_sampling_loc", "grad_attn_weight"),
                       msda_torch.backward(value.double(), shapes, loc.double(), attn.double(), go.double())):
        assert np.abs(t.numpy() - g[name]).max() < 1e-5 * max(1.0, np.abs(g[name]).max()), name
