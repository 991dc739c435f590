"""GPU parity tests of the HIP operators (through the C ABI) against the CPU oracle and the golden
vectors of the real reference.  Tolerances: fp32 re-association only (the kernels compute in fp32)."""
import os

import numpy as np
import pytest
import torch

from oracle import c_ops
from tests import cases
from univs_amd import ops, synth

pytestmark = pytest.mark.gpu


def _msda_gpu(value, shapes, lsi, loc, attn, dev, impl):
    """The standard-layout operator (boundary B2); impl 1 = generic kernel, 2 = LDS-tiled (msda_tiled2.hip: D == 32, P == 4,
    3 <= L <= 4, Lq == S; anything else falls back to the generic kernel)."""
    with ops.configured(msda_impl=impl):
        out = ops.ms_deform_attn_forward(value.to(dev), shapes, lsi, loc.to(dev), attn.to(dev), 128)
        torch.cuda.synchronize()
        ran = ops.msda_last_impl()
        tiled_ok = (value.shape[3] == 32 and loc.shape[4] == 4 and loc.shape[1] == value.shape[1] and loc.shape[3] >= 3)
        assert ran == (2 if (impl == 2 and tiled_ok) else 1), f"impl {impl} requested, {ran} ran"
        assert ops.msda_last_tiled_generation() == (2 if ran == 2 else 0)
    return out.cpu()


def test_g0_kat_float_and_double(cuda, golden_dir):
    """Reference KAT (ops/test.py:35-63) through the reference-shaped entry point, device-resident
    int64 shape tensors included."""
    g = np.load(os.path.join(golden_dir, "g0_msda_kat.npz"))
    shapes = torch.from_numpy(g["shapes"]).to(cuda)
    lsi = torch.from_numpy(g["level_start_index"]).to(cuda)
    for tag, dt in (("double", torch.float64), ("float", torch.float32)):
        v = torch.from_numpy(g[f"value_{tag}"]).to(dt).to(cuda)
        loc = torch.from_numpy(g[f"loc_{tag}"]).to(dt).to(cuda)
        a = torch.from_numpy(g[f"attn_{tag}"]).to(dt).to(cuda)
        out = ops.ms_deform_attn_forward(v, shapes, lsi, loc, a, 2).cpu().numpy()
        ref = g[f"out_{tag}"]
        if tag == "double":
            assert np.allclose(out, ref) and np.abs(out - ref).max() < 1e-15
        else:
            assert np.allclose(out, ref, rtol=1e-2, atol=1e-3) and np.abs(out - ref).max() < 1e-8


@pytest.mark.parametrize("impl", [1, 2], ids=["generic", "tiled2"])
@pytest.mark.parametrize("case", cases.MSDA_CASES, ids=lambda c: c["name"])
def test_msda_matches_oracle_and_golden(cuda, golden_dir, case, impl):
    value, shapes, lsi, loc, attn = cases.msda_inputs(case)
    out = _msda_gpu(value, shapes, lsi, loc, attn, cuda, impl).numpy()
    ref = c_ops.msda_forward(value.numpy(), shapes, lsi, loc.numpy(), attn.numpy())
    err = np.abs(out - ref).max()
    assert err < 2e-5, f"vs oracle: {err}"
    g = np.load(os.path.join(golden_dir, "g1_msda_geometry.npz"))
    sub = cases.msda_query_subset(case, out.shape[1])
    errg = np.abs(out[:, sub] - g[f"{case['name']}/out_subset"]).max()
    assert errg < 2e-5, f"vs reference golden: {errg}"


def _cfg2_inputs(N=2, seed="cfg2"):
    case = dict(name=seed, shapes=[(23, 40), (46, 80), (92, 160)], N=N, M=8, D=32, P=4, encoder=True)
    return cases.msda_inputs(case)


def test_msda_cfg2_size_tiled_equals_generic_and_properties(cuda):
    """BASELINE config 2 geometry (720p: S = 19320).  Size-independent properties: both kernels agree;
    linearity in value; all-zero weights -> exactly zero; a strided subset against the oracle."""
    value, shapes, lsi, loc, attn = _cfg2_inputs()
    o1 = _msda_gpu(value, shapes, lsi, loc, attn, cuda, 1)
    o2 = _msda_gpu(value, shapes, lsi, loc, attn, cuda, 2)
    assert (o1 - o2).abs().max().item() < 2e-5
    v2 = synth.normal("cfg2/value2", tuple(value.shape))
    o_sum = _msda_gpu(value + 2.0 * v2, shapes, lsi, loc, attn, cuda, 2)
    o_b = _msda_gpu(v2, shapes, lsi, loc, attn, cuda, 2)
    assert (o_sum - (o2 + 2.0 * o_b)).abs().max().item() < 1e-4
    o_zero = _msda_gpu(value, shapes, lsi, loc, torch.zeros_like(attn), cuda, 2)
    assert o_zero.abs().max().item() == 0.0
    # oracle on a subset of queries (the oracle is a scalar loop; keep it to seconds)
    sub = torch.arange(0, loc.shape[1], 37)
    ref = c_ops.msda_forward(value.numpy(), shapes, lsi, loc[:, sub].contiguous().numpy(),
                             attn[:, sub].contiguous().numpy())
    assert np.abs(o2[:, sub].numpy() - ref).max() < 2e-5


def test_msda_worst_case_uniform_locations(cuda):
    """U[0,1] sampling locations (SURVEY.md section 8d worst case): nearly every sample misses the staged
    window, so the tiled kernel runs on its global fallback and must still be exact."""
    value, shapes, lsi, loc, attn = _cfg2_inputs(N=1, seed="cfg2u")
    loc = synth.uniform("cfg2u/loc", tuple(loc.shape), -0.05, 1.05)
    o1 = _msda_gpu(value, shapes, lsi, loc, attn, cuda, 1)
    o2 = _msda_gpu(value, shapes, lsi, loc, attn, cuda, 2)
    assert (o1 - o2).abs().max().item() < 2e-5
    # the head-major kernel on the same samples: raw projections that reproduce these locations (their softmax differs
    # from `attn`, so compare with the standard-layout operator fed with ITS locations / weights)
    case = dict(name="cfg2u", shapes=shapes, N=1, M=8, D=32, P=4, encoder=True)
    v2, _, _, proj, n_off, ref = _fused_inputs(case, loc=loc)
    vhm, qhm = ops.msda_pack_head_major(v2.to(cuda), proj.to(cuda), n_off, shapes, 4)
    got = ops.msda_forward_strips(vhm, qhm, ref[:, :, 0].contiguous().to(cuda), shapes, lsi, 8, 4)
    l2, a2 = ops.msda_prepare(proj.to(cuda), n_off, ref.to(cuda), shapes, 8, 3, 4)
    with ops.configured(msda_impl=1):
        want = ops.ms_deform_attn_forward(v2.to(cuda), shapes, lsi, l2, a2)
    assert got is not None and (got - want).abs().max().item() < 3e-5


def _fused_inputs(case, loc=None):
    """Raw projections [N, S, 288-like] + reference points for an encoder-style case: the offsets that reproduce the
    case's sampling locations (or `loc`), random attention logits."""
    value, shapes, lsi, loc0, _ = cases.msda_inputs(case)
    loc = loc0 if loc is None else loc
    N, S, M, D = value.shape
    L, P = len(shapes), case["P"]
    refs = []
    for (h, w) in shapes:
        ys = (torch.arange(h, dtype=torch.float32) + 0.5) / h
        xs = (torch.arange(w, dtype=torch.float32) + 0.5) / w
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        refs.append(torch.stack([xx.reshape(-1), yy.reshape(-1)], -1))
    ref = torch.cat(refs, 0).view(1, S, 1, 2).expand(1, S, L, 2).contiguous()
    norm = torch.tensor([[w, h] for (h, w) in shapes], dtype=torch.float32).view(1, 1, 1, L, 1, 2)
    off = (loc - ref.view(1, S, 1, L, 1, 2)) * norm                        # pixels of the target level
    logits = synth.normal("msda/" + case["name"] + "/logits", (N, S, M * L * P), std=1.5)
    pad = 7                                                                 # columns between the two blocks: n_off is free
    proj = torch.cat([off.reshape(N, S, M * L * P * 2), torch.zeros(N, S, pad), logits], -1).contiguous()
    return value, shapes, lsi, proj, M * L * P * 2 + pad, ref


@pytest.mark.parametrize("case", [c for c in cases.MSDA_CASES if c["encoder"] and c["D"] == 32], ids=lambda c: c["name"])
def test_msda_strips_matches_reference_sequence(cuda, case):
    """Generation 5 (msda_strips.hip: head-major operands, half a head per workgroup) == the reference's sequence softmax /
    reference + offset / normaliser -> ms_deform_attn_forward, evaluated by the oracle, and == our two-operator path on the
    standard layouts; ragged pyramids, four levels (finest first), a single level, far offsets (global fallback)."""
    from oracle import cpu_path
    value, shapes, lsi, proj, n_off, ref = _fused_inputs(case)
    M, L, P = value.shape[2], len(shapes), case["P"]
    want = cpu_path.msda_forward_fused(value, proj, n_off, ref, shapes, lsi, P).numpy()
    vhm, qhm = ops.msda_pack_head_major(value.to(cuda), proj.to(cuda), n_off, shapes, P)
    ref_q = ref[:, :, 0].contiguous().to(cuda)                       # one reference point per query
    assert (ref == ref[:, :, :1]).all()
    got = ops.msda_forward_strips(vhm, qhm, ref_q, shapes, lsi, M, P)
    assert got is not None and ops.msda_last_impl() == 2 and ops.msda_last_tiled_generation() == 5
    loc, attn = ops.msda_prepare(proj.to(cuda), n_off, ref.to(cuda), shapes, M, L, P)
    with ops.configured(msda_impl=1):
        two = ops.ms_deform_attn_forward(value.to(cuda), shapes, lsi, loc, attn)
        assert ops.msda_forward_strips(vhm, qhm, ref_q, shapes, lsi, M, P) is None    # generic forced: the caller's fallback
    with ops.configured(msda_strip_w=8, msda_strip_h=6, msda_grid=7):                  # other tilings, a ragged workgroup split
        alt = ops.msda_forward_strips(vhm, qhm, ref_q, shapes, lsi, M, P)
    assert (alt - got).abs().max().item() < 2e-6
    torch.cuda.synchronize()
    err = np.abs(got.cpu().numpy() - want).max()
    print(f"strips {case['name']}: max abs err vs oracle {err:.2e}, vs generic kernel {(got - two).abs().max().item():.2e}")
    assert err < 3e-5
    assert (got - two).abs().max().item() < 3e-5


def test_msda_strips_cfg2_and_cfg5_size(cuda):
    """Strips at the bench geometries (720p: S = 19 320; 1080p: S = 42 840), 2 frames: == generic kernel on the standard
    layouts; linearity in value; oracle C on every 23rd query."""
    for shapes in ([(23, 40), (46, 80), (92, 160)], [(34, 60), (68, 120), (136, 240)]):
        case = dict(name=f"strips{shapes[0][0]}", shapes=shapes, N=2, M=8, D=32, P=4, encoder=True, far=False)
        value, shapes_, lsi, proj, n_off, ref = _fused_inputs(case)
        M, L, P = 8, 3, 4
        vhm, qhm = ops.msda_pack_head_major(value.to(cuda), proj.to(cuda), n_off, shapes, P)
        ref_q = ref[:, :, 0].contiguous().to(cuda)
        got = ops.msda_forward_strips(vhm, qhm, ref_q, shapes, lsi, M, P)
        assert got is not None and ops.msda_last_tiled_generation() == 5
        loc, attn = ops.msda_prepare(proj.to(cuda), n_off, ref.to(cuda), shapes, M, L, P)
        with ops.configured(msda_impl=1):
            generic = ops.ms_deform_attn_forward(value.to(cuda), shapes, lsi, loc, attn)
        assert (got - generic).abs().max().item() < 3e-5
        got2 = ops.msda_forward_strips(2.0 * vhm, qhm, ref_q, shapes, lsi, M, P)
        assert (got2 - 2.0 * got).abs().max().item() < 1e-5
        sub = np.arange(0, loc.shape[1], 23)
        want = c_ops.msda_forward(value[:1].numpy(), shapes, lsi, loc[:1, sub].contiguous().cpu().numpy(),
                                  attn[:1, sub].contiguous().cpu().numpy())
        assert np.abs(got[:1, sub].cpu().numpy() - want).max() < 3e-5


@pytest.mark.parametrize("case", [c for c in cases.MSDA_CASES if c["encoder"] and c["D"] == 32], ids=lambda c: c["name"])
def test_msda_heads_matches_reference_sequence(cuda, case):
    """Generation 6 (msda_heads.hip: a full head per lane-sample, one workgroup per CU, lockstep column segments) == the
    reference's sequence softmax / reference + offset / normaliser -> ms_deform_attn_forward, evaluated by the oracle, == our
    two-operator path on the standard layouts and == generation 5; ragged pyramids, four levels (finest first), a single
    level, far offsets (global fallback); both scheduling policies, other tilings and ragged grids."""
    from oracle import cpu_path
    value, shapes, lsi, proj, n_off, ref = _fused_inputs(case)
    M, L, P = value.shape[2], len(shapes), case["P"]
    want = cpu_path.msda_forward_fused(value, proj, n_off, ref, shapes, lsi, P).numpy()
    vhm, qhm = ops.msda_pack_heads(value.to(cuda), proj.to(cuda), n_off, shapes, P)
    ref_q = ref[:, :, 0].contiguous().to(cuda)                       # one reference point per query
    got = ops.msda_forward_heads(vhm, qhm, ref_q, shapes, lsi, M, P)
    assert got is not None and ops.msda_last_impl() == 2 and ops.msda_last_tiled_generation() == 6
    loc, attn = ops.msda_prepare(proj.to(cuda), n_off, ref.to(cuda), shapes, M, L, P)
    with ops.configured(msda_impl=1):
        two = ops.ms_deform_attn_forward(value.to(cuda), shapes, lsi, loc, attn)
        assert ops.msda_forward_heads(vhm, qhm, ref_q, shapes, lsi, M, P) is None     # generic forced: the caller's fallback
    for cfg in (dict(msda_sched=1), dict(msda_strip_w=8, msda_strip_h=6, msda_grid=7), dict(msda_grid=24), dict(msda_grid=40, msda_sched=1),
                dict(msda_strip_w=16, msda_strip_h=6)):
        with ops.configured(**cfg):                                                   # schedules / tilings / ragged workgroup splits
            alt = ops.msda_forward_heads(vhm, qhm, ref_q, shapes, lsi, M, P)
        assert alt is not None and (alt - got).abs().max().item() < 2e-6, cfg
    v5, q5 = ops.msda_pack_head_major(value.to(cuda), proj.to(cuda), n_off, shapes, P)
    gen5 = ops.msda_forward_strips(v5, q5, ref_q, shapes, lsi, M, P)
    torch.cuda.synchronize()
    err = np.abs(got.cpu().numpy() - want).max()
    print(f"heads {case['name']}: max abs err vs oracle {err:.2e}, vs generic kernel {(got - two).abs().max().item():.2e}, "
          f"vs generation 5 {(got - gen5).abs().max().item():.2e}")
    assert err < 3e-5
    assert (got - two).abs().max().item() < 3e-5
    assert (got - gen5).abs().max().item() < 1e-5


def test_msda_heads_cfg2_and_cfg5_size(cuda):
    """Generation 6 at the bench geometries (720p: S = 19 320, 5 frames as the clip runs it; 1080p: S = 42 840, 2 frames): ==
    generic kernel on the standard layouts; linearity in value; oracle C on every 23rd query of the first frame; both schedules."""
    for shapes, N in (([(23, 40), (46, 80), (92, 160)], 5), ([(34, 60), (68, 120), (136, 240)], 2)):
        case = dict(name=f"strips{shapes[0][0]}", shapes=shapes, N=N, M=8, D=32, P=4, encoder=True, far=False)
        value, shapes_, lsi, proj, n_off, ref = _fused_inputs(case)
        M, L, P = 8, 3, 4
        vhm, qhm = ops.msda_pack_heads(value.to(cuda), proj.to(cuda), n_off, shapes, P)
        ref_q = ref[:, :, 0].contiguous().to(cuda)
        got = ops.msda_forward_heads(vhm, qhm, ref_q, shapes, lsi, M, P)
        assert got is not None and ops.msda_last_tiled_generation() == 6
        for _ in range(3):      # run to run bit-identical (query slots that repeat a tile's last query compute but do not store)
            assert torch.equal(got, ops.msda_forward_heads(vhm, qhm, ref_q, shapes, lsi, M, P))
        loc, attn = ops.msda_prepare(proj.to(cuda), n_off, ref.to(cuda), shapes, M, L, P)
        with ops.configured(msda_impl=1):
            generic = ops.ms_deform_attn_forward(value.to(cuda), shapes, lsi, loc, attn)
        assert (got - generic).abs().max().item() < 3e-5
        with ops.configured(msda_sched=1):
            assert (ops.msda_forward_heads(vhm, qhm, ref_q, shapes, lsi, M, P) - got).abs().max().item() < 2e-6
        got2 = ops.msda_forward_heads(2.0 * vhm, qhm, ref_q, shapes, lsi, M, P)
        assert (got2 - 2.0 * got).abs().max().item() < 1e-5
        sub = np.arange(0, loc.shape[1], 23)
        want = c_ops.msda_forward(value[:1].numpy(), shapes, lsi, loc[:1, sub].contiguous().cpu().numpy(),
                                  attn[:1, sub].contiguous().cpu().numpy())
        assert np.abs(got[:1, sub].cpu().numpy() - want).max() < 3e-5


def test_msda_heads_uncovered_geometry_and_bad_shapes(cuda):
    vhm = torch.zeros(1, 2, 6, 32, device=cuda)        # M = 2
    ref = torch.zeros(1, 6, 2, device=cuda)
    assert ops.msda_forward_heads(vhm, torch.zeros(1, 2, 6, 3 * 1 * 3, device=cuda), ref, [(2, 3)], [0], 2, 3) is None   # P = 3
    with pytest.raises(RuntimeError):
        ops.msda_forward_heads(vhm, torch.zeros(1, 2, 6, 11, device=cuda), ref, [(2, 3)], [0], 2, 4)                    # row width
    with pytest.raises(RuntimeError):
        ops.msda_forward_heads(vhm, torch.zeros(1, 2, 6, 12, device=cuda), ref, [(2, 3)], [1], 2, 4)                    # level table
    with pytest.raises(RuntimeError):
        ops.msda_forward_heads(torch.zeros(1, 4, 6, 16, device=cuda), torch.zeros(1, 2, 6, 12, device=cuda), ref, [(2, 3)], [0], 2, 4)   # gen-5 layout


@pytest.mark.parametrize("N,S,CB", [(5, 19320, 16), (5, 19320, 32), (5, 19320, 36), (2, 4200, 48), (3, 2352, 16)], ids=str)
@pytest.mark.parametrize("terms", [6, 3], ids=["bf16x6", "f16x3"])
def test_linear_blocked_matches_standard_layout(cuda, linear_terms, terms, N, S, CB):
    """The Linear with the column-blocked epilogue (head-major operands of the strips kernel) == the standard-layout Linear,
    permuted: bit for bit (same arithmetic, different store addresses)."""
    linear_terms(terms)
    K = 256
    Nf = 256 if CB in (16, 32) else 8 * CB
    x = synth.normal("linblk/x", (N, S, K)).to(cuda)
    w = synth.normal("linblk/w", (Nf, K), std=0.05).to(cuda)
    b = synth.normal("linblk/b", (Nf,)).to(cuda)
    std = ops.linear_fused(x, w, b)
    blk = ops.linear_blocked(x, w, b, S, CB)
    if std is None:
        assert blk is None
        return
    assert blk is not None and tuple(blk.shape) == (N, Nf // CB, S, CB)
    assert torch.equal(blk, std.view(N, S, Nf // CB, CB).permute(0, 2, 1, 3).contiguous())


def test_msda_strips_uncovered_geometry_and_bad_shapes(cuda):
    vhm = torch.zeros(1, 4, 6, 16, device=cuda)        # M = 2
    ref = torch.zeros(1, 6, 2, device=cuda)
    assert ops.msda_forward_strips(vhm, torch.zeros(1, 2, 6, 3 * 1 * 3, device=cuda), ref, [(2, 3)], [0], 2, 3) is None   # P = 3
    with pytest.raises(RuntimeError):
        ops.msda_forward_strips(vhm, torch.zeros(1, 2, 6, 11, device=cuda), ref, [(2, 3)], [0], 2, 4)                    # row width
    with pytest.raises(RuntimeError):
        ops.msda_forward_strips(vhm, torch.zeros(1, 2, 6, 12, device=cuda), ref, [(2, 3)], [1], 2, 4)                    # level table


def test_configure_round_trip(cuda):
    """include/univs_hip.h: UnivsConfig -- settings by name, restored on exit; unknown names and bad values raise."""
    base = ops.get_config()
    with ops.configured(msda_impl=1, msda_grid=300):
        c = ops.get_config()
        assert c["msda_impl"] == 1 and c["msda_grid"] == 300 and c["mask_decode_impl"] == base["mask_decode_impl"]
        ops.msda_set_impl(2)
        assert ops.get_config()["msda_impl"] == 2
    assert ops.get_config() == base
    with pytest.raises(KeyError):
        ops.configure(no_such_setting=1)
    with pytest.raises(RuntimeError):
        ops.configure(msda_impl=9)
    assert ops.get_config() == base


def test_msda_fresh_shape_tensors_of_changing_values(cuda):
    """The reference convention (compat drop-in): fresh device tensors for spatial_shapes / level_start_index on
    every call.  The caching allocator hands the same address back for the next resolution's table, so a host-side
    cache keyed on (address, version) would return the previous table; there is no such cache, and an inconsistent
    table raises instead of sampling with the wrong geometry."""
    M, D, P = 2, 32, 4
    for rep in range(6):
        for shapes in ([(8, 6), (4, 3)], [(16, 6), (2, 3)], [(5, 7), (3, 2)]):
            lsi, S = cases.level_start_index(shapes)
            L = len(shapes)
            v = synth.normal(f"fresh/v{S}", (1, S, M, D))
            loc = synth.uniform(f"fresh/l{S}", (1, S, M, L, P, 2), 0.0, 1.0)
            a = torch.softmax(synth.normal(f"fresh/a{S}", (1, S, M, L * P)), -1).view(1, S, M, L, P)
            sh_t = torch.as_tensor(shapes, dtype=torch.long, device=cuda)
            st_t = torch.as_tensor(lsi, dtype=torch.long, device=cuda)
            out = ops.ms_deform_attn_forward(v.to(cuda), sh_t, st_t, loc.to(cuda), a.to(cuda)).cpu().numpy()
            del sh_t, st_t
            ref = c_ops.msda_forward(v.numpy(), shapes, lsi, loc.numpy(), a.numpy())
            assert np.abs(out - ref).max() < 2e-5, (rep, shapes)
    v = torch.zeros(1, 54, 2, 32, device=cuda)
    loc = torch.zeros(1, 54, 2, 2, 4, 2, device=cuda)
    a = torch.zeros(1, 54, 2, 2, 4, device=cuda)
    with pytest.raises(RuntimeError):   # the table of a SMALLER resolution fits inside S but is not this value's table
        ops.ms_deform_attn_forward(v, [(5, 7), (3, 2)], [0, 35], loc, a)
    with pytest.raises(RuntimeError):   # start index that is not the running sum
        ops.ms_deform_attn_forward(v, [(8, 6), (2, 3)], [0, 40], loc, a)


def test_msda_argument_errors(cuda):
    v = torch.zeros(2, 8, 2, 4, device=cuda)
    loc = torch.zeros(2, 3, 2, 1, 2, 2, device=cuda)
    a = torch.zeros(2, 3, 2, 1, 2, device=cuda)
    with pytest.raises(RuntimeError):  # level does not fit S
        ops.ms_deform_attn_forward(v, [(3, 3)], [0], loc, a)
    with pytest.raises(RuntimeError):  # non-contiguous
        ops.ms_deform_attn_forward(v.transpose(2, 3), [(2, 4)], [0], loc, a)
    with pytest.raises(RuntimeError):  # half precision is not dispatched (reference: float/double only)
        ops.ms_deform_attn_forward(v.half(), [(2, 4)], [0], loc.half(), a.half())
    with pytest.raises(RuntimeError):  # grad_output must be [N, Lq, M*D]
        ops.ms_deform_attn_backward(v, [(2, 4)], [0], loc, a, v)
    # empty query set is legal and returns an empty tensor
    out = ops.ms_deform_attn_forward(v, [(2, 4)], [0], loc[:, :0].contiguous(), a[:, :0].contiguous())
    assert tuple(out.shape) == (2, 0, 8)


@pytest.mark.parametrize("case", cases.MASKDEC_CASES, ids=lambda c: c["name"])
def test_mask_decode_matches_oracle(cuda, case):
    e, f = cases.maskdec_inputs(case)
    ops.mask_decode_set_impl(1)
    try:
        out = ops.mask_decode(e.to(cuda), f.to(cuda)).cpu().numpy()
    finally:
        ops.mask_decode_set_impl(0)
    ref = c_ops.mask_decode(e.numpy(), f.numpy())
    # same k-ordered fp32 fmaf chain on both sides -> expected bit-identical; allow 1 ulp-ish slack
    assert np.abs(out - ref).max() < 1e-5
    ref64 = torch.einsum("tqc,tchw->qthw", e.double(), f.double()).numpy()
    assert np.abs(out - ref64).max() < 1e-4


def _bf16x6_runs(case):
    return case["C"] % 64 == 0 and (case["H"] * case["W"]) % 4 == 0


@pytest.mark.parametrize("case", cases.MASKDEC_CASES + cases.MASKDEC_SPLIT_CASES, ids=lambda c: c["name"])
def test_mask_decode_split_bf16_matches_oracle(cuda, case):
    """The "bf16 x 6" kernel (fp32 from an exact 3-way bf16 split of both operands): fp32-level agreement with the
    fp64 contraction and with the oracle's k-ordered fp32 chain; ineligible shapes fall back to the f32 kernel."""
    e, f = cases.maskdec_inputs(case)
    ops.mask_decode_set_impl(2)
    try:
        out = ops.mask_decode(e.to(cuda), f.to(cuda)).cpu().numpy()
        assert ops.mask_decode_last_impl() == (2 if _bf16x6_runs(case) else 1)
    finally:
        ops.mask_decode_set_impl(0)
    ref64 = torch.einsum("tqc,tchw->qthw", e.double(), f.double()).numpy()
    err64 = np.abs(out - ref64).max()
    ref = c_ops.mask_decode(e.numpy(), f.numpy())
    assert err64 < 3e-5 and np.abs(out - ref).max() < 4e-5, (err64, np.abs(out - ref).max())
    # no worse than the fp32 chain itself
    assert err64 <= 2.0 * max(np.abs(ref - ref64).max(), 2e-6)


def test_mask_decode_cfg2_size(cuda):
    """Config-2 size (T=5, Q'=100, 184x320): against torch.einsum on the device + linearity; the default
    dispatch takes the split-bf16 kernel here, and both kernels agree to fp32 rounding."""
    T, Q, C, H, W = 5, 100, 256, 184, 320
    e = synth.normal("md2/e", (T, Q, C), std=0.5).to(cuda)
    f = synth.normal("md2/f", (T, C, H, W), std=0.5).to(cuda)
    out = ops.mask_decode(e, f)
    assert ops.mask_decode_last_impl() == 2
    ref = torch.einsum("tqc,tchw->qthw", e.double(), f.double())
    assert (out.double() - ref).abs().max().item() < 5e-5
    out2 = ops.mask_decode(2.0 * e, f)
    assert (out2 - 2.0 * out).abs().max().item() == 0.0  # scaling by 2 is exact in fp32 (and in the split)
    ops.mask_decode_set_impl(1)
    try:
        out_f32 = ops.mask_decode(e, f)
        assert ops.mask_decode_last_impl() == 1
    finally:
        ops.mask_decode_set_impl(0)
    assert (out_f32.double() - ref).abs().max().item() < 2e-4
    assert (out_f32 - out).abs().max().item() < 5e-5
    # steady-state query count (100 learnable + 10 prompt queries): two row passes
    e110 = synth.normal("md2/e110", (T, 110, C), std=0.5).to(cuda)
    out110 = ops.mask_decode(e110, f)
    assert ops.mask_decode_last_impl() == 2
    assert (out110.double() - torch.einsum("tqc,tchw->qthw", e110.double(), f.double())).abs().max().item() < 5e-5


@pytest.mark.parametrize("impl", [1, 2])
@pytest.mark.parametrize("case", cases.MASKDEC_CASES + cases.MASKDEC_SPLIT_CASES, ids=lambda c: c["name"])
def test_mask_decode_attn_matches_rule(cuda, case, impl):
    e, f = cases.maskdec_inputs(case)
    ops.mask_decode_set_impl(impl)
    try:
        m = ops.mask_decode_attn(e.to(cuda), f.to(cuda)).cpu().numpy()
    finally:
        ops.mask_decode_set_impl(0)
    logits = c_ops.mask_decode(e.numpy(), f.numpy())                      # [Q,T,H,W]
    lg = np.ascontiguousarray(logits.transpose(1, 0, 2, 3)).reshape(case["T"], case["Q"], -1)
    ref = c_ops.attn_mask_from_logits(lg)
    # the f32 kernel evaluates the oracle's own fmaf chain; the split kernel agrees to fp32 rounding
    bad = (m != ref) & (np.abs(lg) > (1e-6 if impl == 1 else 4e-5))
    assert bad.sum() == 0


@pytest.mark.parametrize("T,Q,h,w", [(5, 100, 23, 40), (5, 100, 46, 80), (3, 37, 23, 41), (2, 110, 46, 80)], ids=str)
def test_mask_decode_small_maps_one_shot_is_bit_identical(cuda, T, Q, h, w):
    """The attention masks of the coarse levels (config 2: 23x40 and 46x80, two column tiles per workgroup at 46x80) run the
    one-shot form of the exact-f32 kernel (every k-row of a wave's columns requested at once): same fmaf chain as the chunked
    form, so logits and masks are bit-identical to it (and the masks follow the oracle's rule)."""
    e = synth.normal(f"md1s/e{T}{Q}{h}", (T, Q, 256), std=0.5).to(cuda)
    f = synth.normal(f"md1s/f{T}{Q}{h}", (T, 256, h, w), std=0.5).to(cuda)
    with ops.configured(mask_decode_impl=1):
        lg1 = ops.mask_decode(e, f)
        m1 = ops.mask_decode_attn(e, f)
        with ops.configured(mask_decode_chunked=1):
            lg0 = ops.mask_decode(e, f)
            m0 = ops.mask_decode_attn(e, f)
    m_default = ops.mask_decode_attn(e, f)                      # by size: the exact-f32 kernel at these sizes
    assert ops.mask_decode_last_impl() == 1
    assert torch.equal(lg1, lg0) and torch.equal(m1, m0) and torch.equal(m_default, m1)
    ref = torch.einsum("tqc,tchw->qthw", e.double(), f.double())
    assert (lg1.double() - ref).abs().max().item() < 1e-4
    want = c_ops.attn_mask_from_logits(np.ascontiguousarray(lg1.cpu().numpy().transpose(1, 0, 2, 3)).reshape(T, Q, -1))
    assert np.array_equal(m1.cpu().numpy(), want)


def test_mask_decode_attn_row_reset(cuda):
    T, Q, C, h, w = 1, 3, 64, 4, 5
    f = torch.ones(T, C, h, w, device=cuda)
    e = torch.zeros(T, Q, C, device=cuda)
    e[0, 0] = -1.0      # every logit negative -> fully masked row -> reset to all-visible
    e[0, 1] = 1.0       # every logit positive -> nothing masked
    e[0, 2, 0] = -1.0   # negative as well
    m = ops.mask_decode_attn(e, f).cpu()
    assert m.sum().item() == 0
    f[0, :, 0, 0] = -1.0  # one key flips sign for rows 0 and 2 -> that key visible, the rest masked
    m = ops.mask_decode_attn(e, f).cpu()
    assert m[0, 0].tolist() == [False] + [True] * 19
    assert m[0, 1].tolist() == [True] + [False] * 19


@pytest.mark.parametrize("case", cases.WINATTN_CASES, ids=lambda c: c["name"])
def test_window_attention_matches_oracle_and_golden(cuda, golden_dir, case):
    x, mask = cases.winattn_inputs(case)
    dim, nH, win = case["dim"], case["heads"], case["win"]
    ntok = win * win
    p = case["name"] + "."
    w_qkv = synth.make_param(p + "qkv.weight", (3 * dim, dim)); b_qkv = synth.make_param(p + "qkv.bias", (3 * dim,))
    w_proj = synth.make_param(p + "proj.weight", (dim, dim)); b_proj = synth.make_param(p + "proj.bias", (dim,))
    table = synth.make_param(p + "relative_position_bias_table", ((2 * win - 1) ** 2, nH))
    ch, cw = torch.meshgrid(torch.arange(win), torch.arange(win), indexing="ij")
    coords = torch.stack([ch.reshape(-1), cw.reshape(-1)])
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0) + (win - 1)
    index = rel[..., 0] * (2 * win - 1) + rel[..., 1]
    bias = table[index.reshape(-1)].view(ntok, ntok, nH).permute(2, 0, 1).contiguous()
    qkv = (x @ w_qkv.t() + b_qkv).view(x.shape[0], ntok, 3, nH, dim // nH).contiguous()
    scale = (dim // nH) ** -0.5
    core = ops.window_attention(qkv.to(cuda), bias.to(cuda), None if mask is None else mask.to(cuda),
                                case["nW"], scale).cpu()
    ref = torch.from_numpy(c_ops.window_attention(qkv.numpy(), bias.numpy(),
                                                  None if mask is None else mask.numpy(), scale))
    assert (core - ref).abs().max().item() < 2e-5
    y = core @ w_proj.t() + b_proj
    g = np.load(os.path.join(golden_dir, "g_window_attention.npz"))
    assert (y - torch.from_numpy(g[f"{case['name']}/out"])).abs().max().item() < 5e-5


@pytest.mark.parametrize("shape,size", [((3, 5, 184, 320), (92, 160)), ((2, 7, 46, 80), (23, 40)), ((2, 3, 16, 24), (64, 96)),
                                        ((1, 4, 37, 53), (19, 31)), ((1, 2, 8, 12), (8, 12)), ((2, 2, 5, 7), (1, 1))],
                         ids=lambda v: "x".join(map(str, v)))
def test_bilinear_resample_matches_torch(cuda, shape, size):
    """ops.bilinear_resample == F.interpolate(bilinear, align_corners=False): down / up / odd sizes / identity."""
    x = synth.normal("resample/" + "x".join(map(str, shape)), shape)
    ref = torch.nn.functional.interpolate(x, size=size, mode="bilinear", align_corners=False)
    got = ops.bilinear_resample(x.to(cuda), size).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item())
    add = synth.normal("resample/add/" + "x".join(map(str, shape)), tuple(ref.shape))
    got2 = ops.bilinear_resample(x.to(cuda), size, addend=add.to(cuda)).cpu()
    assert (got2 - (add + ref)).abs().max().item() < 4e-6 * max(1.0, ref.abs().max().item())
    xt = x.transpose(0, 1)                      # non-contiguous inputs are copied once
    assert torch.equal(ops.bilinear_resample(xt.to(cuda), size).cpu(), got.transpose(0, 1))
    with pytest.raises(RuntimeError):
        ops.bilinear_resample(x, size)          # CPU tensors are refused (no fallback)


@pytest.mark.parametrize("shape", [(5, 256, 92, 160), (2, 32, 23, 40), (1, 64, 7, 130), (3, 32, 1, 2), (1, 32, 40, 66)], ids=lambda v: str(v))
def test_upsample2x_add_with_group_norm(cuda, shape):
    """ops.upsample2x_add == bilinear_resample(x, 2x, addend) bit for bit (same taps, weights and expression; the data movement
    differs: 8-byte loads + lane exchanges), incl. the clamped border columns / rows and widths where column groups of different
    rows share a wave; with `affine` = group_norm_affine(addend) == group_norm(addend) + upsample bit for bit, and within fp32
    rounding of F.group_norm + F.interpolate (the FPN top-down step of msdeformattn.py:349-351)."""
    F = torch.nn.functional
    N, C, H, W = shape
    x = synth.normal("up2/x/" + "x".join(map(str, shape)), shape)
    add = synth.normal("up2/a/" + "x".join(map(str, shape)), (N, C, 2 * H, 2 * W)) * 2.0 + 0.3
    g_ = 1.0 + 0.2 * synth.normal(f"up2/g/{C}", (C,))
    b_ = 0.1 * synth.normal(f"up2/b/{C}", (C,))
    xd, ad, gd, bd = x.to(cuda), add.to(cuda), g_.to(cuda), b_.to(cuda)
    y = ops.upsample2x_add(xd, ad)
    assert y is not None and torch.equal(y, ops.bilinear_resample(xd, (2 * H, 2 * W), addend=ad))
    ref = add + F.interpolate(x, size=(2 * H, 2 * W), mode="bilinear", align_corners=False)
    assert (y.cpu() - ref).abs().max().item() < 4e-6 * max(1.0, ref.abs().max().item())
    aff = ops.group_norm_affine(ad, 32, gd, bd, 1e-5)
    assert tuple(aff.shape) == (N * C, 2)
    y2 = ops.upsample2x_add(xd, ad, aff)
    two = ops.bilinear_resample(xd, (2 * H, 2 * W), addend=ops.group_norm(ad, 32, gd, bd, 1e-5))
    assert torch.equal(y2, two)
    ref2 = F.group_norm(add.double(), 32, g_.double(), b_.double(), 1e-5) + F.interpolate(x.double(), size=(2 * H, 2 * W), mode="bilinear", align_corners=False)
    ref2_32 = F.group_norm(add, 32, g_, b_, 1e-5) + F.interpolate(x, size=(2 * H, 2 * W), mode="bilinear", align_corners=False)
    err, err32 = (y2.cpu().double() - ref2).abs().max().item(), (ref2_32.double() - ref2).abs().max().item()
    assert err < max(4.0 * err32, 2e-5), (err, err32)          # (GroupNorm's own bound: test_group_norm_matches_torch)
    assert ops.upsample2x_add(xd, ad[..., :-1]) is None and ops.upsample2x_add(xd[..., :-1], ad[..., :-2]) is None   # not 2x / odd width


@pytest.mark.parametrize("rows,C", [(1000, 96), (513, 192), (300, 256), (257, 384), (129, 768), (65, 1536), (33, 3072), (7, 8), (0, 96), (77, 640), (50, 64)],
                         ids=lambda v: str(v))
def test_layer_norm_matches_torch(cuda, rows, C):
    """ops.layer_norm == F.layer_norm(x [+ residual]) for every row length of the path (Swin 96..768, patch merging
    384..3072, encoder / decoder 256), with and without the fused residual and the returned sum."""
    x = synth.normal(f"ln/x/{rows}/{C}", (rows, C)) * 3.0 + 0.5
    r = synth.normal(f"ln/r/{rows}/{C}", (rows, C))
    w = 1.0 + 0.1 * synth.uniform(f"ln/w/{C}", (C,))
    b = 0.05 * synth.uniform(f"ln/b/{C}", (C,))
    F = torch.nn.functional
    xd, rd, wd, bd = (t.to(cuda) for t in (x, r, w, b))
    got = ops.layer_norm(xd, wd, bd, 1e-5).cpu()
    assert (got - F.layer_norm(x, (C,), w, b, 1e-5)).abs().max().item() < 2e-5 if rows else got.shape == x.shape
    s, got2 = ops.layer_norm(xd, wd, bd, 1e-5, residual=rd, return_sum=True)
    if rows:
        assert torch.equal(s.cpu(), x + r)
        assert (got2.cpu() - F.layer_norm(x + r, (C,), w, b, 1e-5)).abs().max().item() < 2e-5
        got3 = ops.layer_norm(xd.view(1, rows, C), wd, bd, 1e-5, residual=rd.view(1, rows, C))
        assert torch.equal(got3.view(rows, C), got2)


def test_layer_norm_argument_errors(cuda):
    x = torch.zeros(4, 10, device=cuda)
    w = torch.ones(10, device=cuda)
    with pytest.raises(RuntimeError):
        ops.layer_norm(x, w, w)                      # C % 4 != 0: not covered, loud
    with pytest.raises(RuntimeError):
        ops.layer_norm(x.cpu(), w.cpu(), w.cpu())    # no CPU fallback
    with pytest.raises(RuntimeError):
        ops.layer_norm(torch.zeros(4, 8, device=cuda), w, w)


@pytest.mark.parametrize("case", cases.MSDA_BWD_CASES, ids=lambda c: c["name"])
def test_msda_backward_matches_autograd_oracle(cuda, case):
    """univs_msda_backward_f32 against autograd through the oracle's differentiable restatement of
    ms_deform_attn_core_pytorch (oracle/msda_torch.py), the construction the reference's own
    ops/test.py:check_gradient_numerical uses."""
    from oracle import msda_torch
    value, shapes, lsi, loc, attn = cases.msda_inputs(case)
    go = synth.normal("msda_bwd/go/" + case["name"], (value.shape[0], loc.shape[1], value.shape[2] * value.shape[3]))
    ref = msda_torch.backward(value.double(), shapes, loc.double(), attn.double(), go.double())
    got = ops.ms_deform_attn_backward(value.to(cuda), shapes, lsi, loc.to(cuda), attn.to(cuda), go.to(cuda))
    for name, g, r in zip(("grad_value", "grad_sampling_loc", "grad_attn_weight"), got, ref):
        scale = max(1.0, r.abs().max().item())
        assert (g.cpu().double() - r).abs().max().item() < 2e-4 * scale, name


def test_msda_backward_matches_reference_golden(cuda, golden_dir):
    """... and against gradients of the REAL reference's core (tests/golden/g13_msda_backward.npz)."""
    g = np.load(os.path.join(golden_dir, "g13_msda_backward.npz"))
    case = cases.MSDA_BWD_CASE
    value, shapes, lsi, loc, attn = cases.msda_inputs(case)
    go = synth.normal("msda_bwd/go/" + case["name"], (value.shape[0], loc.shape[1], value.shape[2] * value.shape[3]))
    got = ops.ms_deform_attn_backward(value.to(cuda), shapes, lsi, loc.to(cuda), attn.to(cuda), go.to(cuda))
    for name, t in zip(("grad_value", "grad_sampling_loc", "grad_attn_weight"), got):
        r = g[name]
        assert np.abs(t.cpu().numpy() - r).max() < 2e-4 * max(1.0, np.abs(r).max()), name


@pytest.mark.parametrize("shape,groups,relu", [((5, 256, 23, 40), 32, False), ((2, 256, 46, 80), 32, True), ((1, 64, 184, 320), 32, True),
                                               ((3, 32, 7, 9), 8, False), ((2, 16, 1, 1), 4, True)], ids=lambda v: str(v))
def test_group_norm_matches_torch(cuda, shape, groups, relu):
    """ops.group_norm == F.group_norm [+ relu]: pixel-decoder sizes, odd planes (scalar path), 1x1 planes."""
    x = synth.normal("gn/x/" + "x".join(map(str, shape)), shape) * 2.0 + 0.7
    w = 1.0 + 0.1 * synth.uniform(f"gn/w/{shape[1]}", (shape[1],))
    b = 0.05 * synth.uniform(f"gn/b/{shape[1]}", (shape[1],))
    ref = torch.nn.functional.group_norm(x, groups, w, b, 1e-5)
    ref = torch.relu(ref) if relu else ref
    got = ops.group_norm(x.to(cuda), groups, w.to(cuda), b.to(cuda), 1e-5, relu=relu).cpu()
    assert (got - ref).abs().max().item() < 2e-5
    with pytest.raises(RuntimeError):
        ops.group_norm(x, groups, w, b)


@pytest.mark.parametrize("N,h,L,S", [(2, 8, 13, 920), (1, 8, 100, 3680), (1, 2, 5, 14720), (1, 1, 3, 20000), (3, 4, 7, 33), (1, 30, 77, 77)],
                         ids=lambda v: str(v))
def test_masked_softmax_matches_torch(cuda, N, h, L, S):
    """ops.masked_softmax_ == masked_fill(-inf) + softmax, every register-resident size class and the streaming one."""
    x = synth.normal(f"sm/x/{N}/{h}/{L}/{S}", (N, h, L, S)) * 3.0
    m = synth.uniform(f"sm/m/{N}/{L}/{S}", (N, L, S)) > 0.3
    m[..., 0] = False                       # no fully masked row (the caller guarantees that, ...decoder_univs.py:390)
    ref = torch.softmax(x.masked_fill(m.unsqueeze(1), float("-inf")), dim=-1)
    xd = x.to(cuda)
    got = ops.masked_softmax_(xd, m.to(cuda))
    assert got.data_ptr() == xd.data_ptr()
    assert (got.cpu() - ref).abs().max().item() < 4e-6          # probabilities <= 1: a few ulp
    got2 = ops.masked_softmax_(x.to(cuda), None).cpu()
    assert (got2 - torch.softmax(x, dim=-1)).abs().max().item() < 4e-6
    got3 = ops.masked_softmax_(x.to(cuda), m.to(cuda).to(torch.uint8)).cpu()
    assert torch.equal(got3, got.cpu())


def _window_image_inputs(B, H, W, ws, shift, nH):
    hd, n = 32, ws * ws
    tag = f"wai/{B}/{H}/{W}/{ws}/{shift}/{nH}"
    qkv = synth.normal(tag + "/qkv", (B, H * W, 3, nH, hd))
    qb = synth.normal(tag + "/qb", (3 * nH * hd,)) * 0.5
    bias = synth.normal(tag + "/bias", (nH, n, n))
    Hp, Wp = (H + ws - 1) // ws * ws, (W + ws - 1) // ws * ws
    nW = (Hp // ws) * (Wp // ws)
    mask = None
    if shift:
        # the reference's shift mask (swin.py:413-440): region ids on the padded canvas -> 0 / -100
        img = torch.zeros(1, Hp, Wp, 1)
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                img[:, hs, wsl, :] = cnt
                cnt += 1
        mw = img.view(1, Hp // ws, ws, Wp // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, n)
        am = mw.unsqueeze(1) - mw.unsqueeze(2)
        mask = am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)
        assert mask.shape[0] == nW
    return qkv, qb, bias, mask


@pytest.mark.parametrize("B,H,W,ws,shift,nH", [(2, 14, 21, 7, 0, 3), (2, 14, 21, 7, 3, 3), (1, 23, 40, 7, 3, 2), (1, 24, 36, 12, 6, 4),
                                                (2, 9, 11, 12, 0, 1), (1, 7, 7, 7, 0, 2), (1, 20, 31, 9, 4, 2), (3, 30, 45, 7, 3, 6)],
                         ids=lambda v: str(v))
@pytest.mark.parametrize("mma", ["f32", "f16x3"])
def test_window_attention_image_matches_reference_data_movement(cuda, mma, B, H, W, ws, shift, nH):
    """Image-mode window attention == pad -> roll -> window_partition -> core -> window_reverse -> roll -> crop
    (the oracle's restatement of swin.py:252-284 around the already-pinned core), incl. padded pixels (qkv = bias),
    shifted windows with the 0/-100 mask, windows larger than the image, and both window sizes."""
    from oracle import cpu_path
    qkv, qb, bias, mask = _window_image_inputs(B, H, W, ws, shift, nH)
    ref = cpu_path.window_attention_image(qkv, qb, bias, mask, H, W, ws, shift, 32 ** -0.5)
    got = ops.window_attention_image(qkv.to(cuda), qb.to(cuda), bias.to(cuda), mask.to(cuda) if mask is not None else None,
                                     H, W, ws, shift, 32 ** -0.5, mma=mma).cpu()
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item()
    print(f"window attention {mma} {B, H, W, ws, shift, nH}: max-abs-err {err:.2e}")
    assert err < (4e-6 if mma == "f16x3" else 2e-5)


def _xattn_reference(q, k, v, mask, H, scale):
    """nn.MultiheadAttention's core in fp64: q [L, N, E], k / v [S, N, E], mask [N, L, S] bool (True = masked)."""
    L, N, E = q.shape
    S = k.shape[0]
    d = E // H
    qd = q.double().reshape(L, N, H, d).permute(1, 2, 0, 3)
    kd = k.double().reshape(S, N, H, d).permute(1, 2, 0, 3)
    vd = v.double().reshape(S, N, H, d).permute(1, 2, 0, 3)
    sc = torch.matmul(qd * scale, kd.transpose(-1, -2))
    if mask is not None:
        sc = sc.masked_fill(mask.bool()[:, None], float("-inf"))
    out = torch.matmul(torch.softmax(sc, -1), vd)
    return out.permute(2, 0, 1, 3).reshape(L, N, E)


@pytest.mark.parametrize("L,S,N,H,masked", [(100, 920, 5, 8, True), (100, 14720, 2, 8, True), (20, 3680, 3, 8, True), (100, 3680, 5, 8, False),
                                             (7, 1000, 1, 2, True), (130, 1504, 2, 4, True), (112, 516, 1, 8, True),
                                             (500, 500, 1, 8, True), (2000, 2000, 1, 8, True), (300, 260, 2, 4, False),
                                             (550, 550, 1, 8, True), (100, 701, 2, 4, True), (64, 67, 3, 2, True), (110, 110, 5, 8, False)],
                         ids=lambda v: str(v))
def test_cross_attention_matches_torch(cuda, L, S, N, H, masked):
    """ops.cross_attention (csrc/cross_attn.hip: scores, mask, softmax and P V in one pass over the keys, three-product fp16
    arithmetic, per-segment partials merged by a second kernel) == nn.MultiheadAttention's core
    (transformer_layers.py:95-115) in fp64 to fp32 rounding; masks with whole 32-key blocks and whole segments masked for
    some queries, a query with one visible key, more than 128 queries (chunks), S not a multiple of 32, S not a multiple of 4 (the mask
    rows are padded to whole dwords: 550 = the prompted clip's 110 queries x 5 frames)."""
    E = 32 * H
    q = synth.normal(f"xa/q/{L}x{N}x{E}", (L, N, E))
    k = synth.normal(f"xa/k/{S}x{N}x{E}", (S, N, E))
    v = synth.normal(f"xa/v/{S}x{N}x{E}", (S, N, E))
    mask = None
    if masked:
        g = torch.Generator().manual_seed(L * 7 + S)
        mask = torch.rand(N, L, S, generator=g) < 0.6
        mask[:, 1::5, : S // 2] = True                      # half of the keys (whole segments) masked for some queries
        mask[:, 2::7, 64:640] = True                        # runs of whole 32-key blocks
        mask[0, 3] = True
        mask[0, 3, S - 5] = False                            # one visible key
        mask[..., 0] = mask[..., 0] & ~mask.all(-1)          # no fully masked row: its first key becomes visible
        assert not mask.all(-1).any()
    scale = 32 ** -0.5
    dev = [t.to(cuda) if t is not None else None for t in (q, k, v, mask)]
    got = ops.cross_attention(dev[0], dev[1], dev[2], dev[3], H, scale)
    assert got is not None and tuple(got.shape) == (L, N, E)
    ref64 = _xattn_reference(dev[0], dev[1], dev[2], dev[3], H, scale)
    sc32 = torch.matmul((dev[0] * scale).reshape(L, N, H, 32).permute(1, 2, 0, 3), dev[1].reshape(S, N, H, 32).permute(1, 2, 3, 0))
    if mask is not None:
        sc32 = sc32.masked_fill(dev[3][:, None], float("-inf"))
    ref32 = torch.matmul(torch.softmax(sc32, -1), dev[2].reshape(S, N, H, 32).permute(1, 2, 0, 3)).permute(2, 0, 1, 3).reshape(L, N, E)
    err = (got.double() - ref64).abs().max().item()
    err32 = (ref32.double() - ref64).abs().max().item()
    print(f"cross attention {L, S, N, H, masked}: max-abs-err {err:.2e} (ATen fp32 path {err32:.2e})")
    assert torch.isfinite(got).all()
    assert err < max(4.0 * err32, 5e-6), (err, err32)
    if mask is not None:                                     # uint8 masks are taken as they are
        assert torch.equal(ops.cross_attention(dev[0], dev[1], dev[2], dev[3].to(torch.uint8), H, scale), got)


def test_cross_attention_ranges_and_module_path(cuda):
    """Operands outside fp16's range (wave-uniform power-of-two scaling inside the kernel), large score spreads (the lazy
    reference maximum is raised many times), and layers.MultiheadAttention taking the fused core == the unfused path."""
    from univs_amd import layers
    from univs_amd.switches import override
    L, S, N, H = 100, 3680, 2, 8
    E = 32 * H
    q = synth.normal("xa2/q", (L, N, E))
    k = synth.normal("xa2/k", (S, N, E))
    v = synth.normal("xa2/v", (S, N, E))
    scale = 32 ** -0.5
    # small v: a block whose |v| < 2^-4 throughout is range-scaled UP; the scale is undone in fp32 on the block's P V' (ADVICE r04:
    # undoing it on the probabilities flushed them to fp16 subnormals -- 10 % error at |v| ~ 1e-2, zeros at 1e-5)
    for name, sq, sk, sv, tol in (("large v", 1.0, 1.0, 1.0e6, 2e-5), ("small v 1e-2", 1.0, 1.0, 1.0e-2, 2e-5), ("small v 1e-3", 1.0, 1.0, 1.0e-3, 2e-5),
                                  ("small v 1e-6", 1.0, 1.0, 1.0e-6, 2e-5), ("large k, small q", 1.0e-5, 1.0e5, 1.0, 2e-5),
                                  ("large q, small k", 3.0e5, 2.0e-6, 1.0, 2e-5), ("sharp scores", 6.0, 6.0, 1.0, 1e-4)):
        a = [(q * sq).to(cuda), (k * sk).to(cuda), (v * sv).to(cuda)]
        got = ops.cross_attention(a[0], a[1], a[2], None, H, scale)
        ref = _xattn_reference(a[0], a[1], a[2], None, H, scale)
        err = ((got.double() - ref).abs().max() / ref.abs().max()).item()
        print(f"cross attention, {name}: relative max error {err:.2e}")
        assert torch.isfinite(got).all() and err < tol, (name, err)
    # v with scaled and unscaled 32-key blocks side by side in every segment (magnitude by key block)
    blk = (torch.arange(S) // 32) % 3
    vmix = v * torch.tensor([1.0, 1.0e-3, 3.0e-6])[blk].view(S, 1, 1)
    got = ops.cross_attention(q.to(cuda), k.to(cuda), vmix.to(cuda), None, H, scale)
    ref = _xattn_reference(q.to(cuda), k.to(cuda), vmix.to(cuda), None, H, scale)
    err = ((got.double() - ref).abs().max() / ref.abs().max()).item()
    print(f"cross attention, mixed v blocks: relative max error {err:.2e}")
    assert err < 2e-5, err
    # k / v as column slices of wider projections (one Linear for the keys of several layers): same result as dense copies
    wide_k = torch.cat([synth.normal("xa2/wk0", (S, N, E)), k, synth.normal("xa2/wk2", (S, N, E))], -1).to(cuda)
    wide_v = torch.cat([v, synth.normal("xa2/wv1", (S, N, 2 * E))], -1).to(cuda)
    msk2 = (torch.rand(N, L, S, generator=torch.Generator().manual_seed(9)) < 0.7).to(cuda)
    dense = ops.cross_attention(q.to(cuda), k.to(cuda), v.to(cuda), msk2, H, scale)
    sliced = ops.cross_attention(q.to(cuda), wide_k[..., E:2 * E], wide_v[..., :E], msk2, H, scale)
    assert torch.equal(dense, sliced)
    mha = layers.MultiheadAttention(E, H).to(cuda).eval()
    with torch.no_grad():
        for n_, p_ in mha.named_parameters():
            p_.copy_(synth.normal(f"xa2/mha/{n_}", tuple(p_.shape), std=0.08).to(cuda))
        tgt = synth.normal("xa2/tgt", (L, N, E)).to(cuda)
        mem = synth.normal("xa2/mem", (S, N, E)).to(cuda)
        key = mem + synth.normal("xa2/pos", (S, N, E)).to(cuda)
        msk = (torch.rand(N, L, S, generator=torch.Generator().manual_seed(5)) < 0.5).to(cuda)
        fused = mha(tgt, key, mem, attn_mask=msk)[0]
        with override(fused_cross_attention=False):
            plain = mha(tgt, key, mem, attn_mask=msk)[0]
        # precomputed key / value projections (the decoder's per-level batching) == the module's own projections
        kk = layers.linear(key, mha.in_proj_weight[E:2 * E], mha.in_proj_bias[E:2 * E])
        vv = layers.linear(mem, mha.in_proj_weight[2 * E:], mha.in_proj_bias[2 * E:])
        given = mha(tgt, None, None, attn_mask=msk, kv=(kk, vv))[0]
    assert (fused - plain).abs().max().item() < 2e-5
    assert torch.equal(given, fused)
    # the decoder's spatio-temporal self-attention (...decoder_univs.py:408-414): one batch entry, Q' T tokens, a [L, S] mask,
    # q and k from ONE projection of tgt + pos, v from tgt
    for Ls in (500, 550):                                    # (550: rows that are not whole dwords -- padded once per mask object)
        with torch.no_grad():
            t1 = synth.normal(f"xa2/sa/tgt{Ls}", (Ls, 1, E)).to(cuda)
            qk = t1 + synth.normal(f"xa2/sa/pos{Ls}", (Ls, 1, E)).to(cuda)
            m2 = (torch.rand(Ls, Ls, generator=torch.Generator().manual_seed(6)) < 0.4).to(cuda)
            m2[torch.arange(Ls), torch.arange(Ls)] = False
            for mm in (m2, None):
                sa_fused = mha(qk, qk, t1, attn_mask=mm)[0]
                sa_again = mha(qk, qk, t1, attn_mask=mm)[0]
                with override(fused_cross_attention=False):
                    sa_plain = mha(qk, qk, t1, attn_mask=mm)[0]
                assert (sa_fused - sa_plain).abs().max().item() < 2e-5 and torch.equal(sa_fused, sa_again)
            if Ls % 4:
                assert ops.pad4_mask(m2) is ops.pad4_mask(m2) and ops.pad4_mask(m2).shape == (Ls, (Ls + 3) // 4 * 4)
                m2[0, 1] = ~m2[0, 1]                             # an in-place change: a new padded copy
                assert bool(ops.pad4_mask(m2)[0, 1]) == bool(m2[0, 1])
    assert ops.cross_attention(torch.zeros(4, 1, 64, device=cuda), torch.zeros(16, 1, 64, device=cuda), torch.zeros(16, 1, 64, device=cuda),
                               None, 2, 1.0) is None                                            # fewer than 32 keys
    assert ops.cross_attention(torch.zeros(4, 1, 64), torch.zeros(64, 1, 64), torch.zeros(64, 1, 64), None, 2, 1.0) is None   # CPU


@pytest.mark.parametrize("B,H,W,ws,shift,nH", [(2, 14, 21, 7, 3, 3), (1, 20, 31, 9, 4, 2), (1, 23, 40, 7, 0, 2)], ids=lambda v: str(v))
def test_window_attention_f16x3_out_of_range_operands(cuda, B, H, W, ws, shift, nH):
    """The three-product window attention (the Swin default) on operands OUTSIDE fp16's range: windows whose k, v or scaled q
    hold magnitudes >= 2^15 are brought into range by a wave-uniform power of two inside the kernel (undone on the fp32
    scores / the normaliser), so the result stays that of the exact-f32 kernel instead of Inf - Inf = NaN -- large v (1e6),
    large k with small q, large q with small k, and everything large but scores still finite in fp32."""
    qkv, qb, bias, mask = _window_image_inputs(B, H, W, ws, shift, nH)
    args = (H, W, ws, shift, 32 ** -0.5)
    dev = lambda t: t.to(cuda) if t is not None else None  # noqa: E731
    base = [dev(t) for t in (qkv, qb, bias, mask)]
    ok = ops.window_attention_image(*base, *args, mma="f16x3")
    assert torch.isfinite(ok).all()
    # (tolerance of "k and v beyond 65504": its scores are ~700 x larger than usual, ~2 000 in the exp2 domain, where fp32's own
    # resolution of a score is 1.2e-4 -- both kernels carry that error against the exact result)
    # small v: a block whose |v| < 2^-4 throughout is range-scaled UP; the scale is undone in fp32 on the block's P V' (ADVICE r04:
    # undoing it on the probabilities flushed them to fp16 subnormals -- 10 % error at |v| ~ 1e-2, zeros at 1e-5)
    for name, sq, sk, sv, tol in (("large v", 1.0, 1.0, 1.0e6, 2e-5), ("small v 1e-2", 1.0, 1.0, 1.0e-2, 2e-5), ("small v 1e-3", 1.0, 1.0, 1.0e-3, 2e-5),
                                  ("small v 1e-6", 1.0, 1.0, 1.0e-6, 2e-5), ("large k, small q", 1.0e-5, 1.0e5, 1.0, 2e-5),
                                  ("large q, small k", 3.0e5, 2.0e-6, 1.0, 2e-5), ("k and v beyond 65504", 1.0e-2, 7.0e4, 7.0e4, 5e-4),
                                  ("one huge channel", 1.0, 1.0, 1.0, 2e-5)):
        q2 = qkv.clone()
        q2[:, :, 0] *= sq
        q2[:, :, 1] *= sk
        q2[:, :, 2] *= sv
        qb2 = qb.clone().view(3, -1)
        qb2[0] *= sq
        qb2[1] *= sk
        qb2[2] *= sv
        if name == "one huge channel":       # a single outlier in one token of one window: only that window is rescaled
            q2[0, 5, 1, 0, 3] = 9.0e4
        a2 = [dev(q2), dev(qb2.reshape(-1)), base[2], base[3]]
        got = ops.window_attention_image(*a2, *args, mma="f16x3")
        ref = ops.window_attention_image(*a2, *args, mma="f32")
        assert torch.isfinite(got).all(), name
        scale = ref.abs().max().item()
        err = (got - ref).abs().max().item() / scale
        print(f"f16x3 window attention, {name} {B, H, W, ws, shift, nH}: relative max error {err:.2e}")
        assert err < tol, (name, err)
        if name == "one huge channel":       # windows without the outliers are untouched (bit-identical to the plain run)
            assert (got == ok).float().mean().item() > 0.5


# Tolerance of the fp16-operand window attention (UNIVS_MMA_F16) against the fp32 operator, on unit-normal q, k, v, bias:
# each operand carries a relative rounding error of 2^-11, which over 32 channels and up to 144 keys gives errors of a few
# 1e-4 of the output scale; 4e-3 absolute leaves a factor ~4 over the largest error measured on these cases.
WINDOW_F16_ATOL = 4e-3
# against the restatement that rounds at the SAME points (oracle/cpu_path.py: window_attention(mma="f16")) only the
# summation order and the last bit of exp2 differ; an exp2 off by one ulp can flip the fp16 rounding of ONE probability p,
# which moves the output by 2^-11 p |v| (2.1e-4 measured; |v| reaches 4 on these inputs)
WINDOW_F16_ATOL_SAME_ROUNDING = 5e-4


@pytest.mark.parametrize("B,H,W,ws,shift,nH", [(2, 14, 21, 7, 0, 3), (2, 14, 21, 7, 3, 3), (1, 23, 40, 7, 3, 2), (1, 24, 36, 12, 6, 4),
                                                (2, 9, 11, 12, 0, 1), (1, 7, 7, 7, 0, 2), (1, 20, 31, 9, 4, 2), (1, 13, 26, 8, 0, 3),
                                                (2, 68, 120, 12, 6, 6)], ids=lambda v: str(v))
def test_window_attention_fp16_operands(cuda, B, H, W, ws, shift, nH):
    """ops.window_attention_image(mma="f16") (BASELINE config 5's fp16 MFMA window attention) -- against the oracle's
    restatement with the same rounding points (tight: layout, padding, masks, every tile size 4 / 6 / 9 blocks) and
    against the fp32 operator (the stated tolerance of the variant)."""
    from oracle import cpu_path
    qkv, qb, bias, mask = _window_image_inputs(B, H, W, ws, shift, nH)
    args = (H, W, ws, shift, 32 ** -0.5)
    dev = [t.to(cuda) if t is not None else None for t in (qkv, qb, bias, mask)]
    got = ops.window_attention_image(*dev, *args, mma="f16")
    f32 = ops.window_attention_image(*dev, *args, mma="f32")
    assert torch.equal(f32, ops.window_attention_image(*dev, *args))            # "f32" is the default operator
    same = cpu_path.window_attention_image(qkv, qb, bias, mask, *args, mma="f16")
    e_same = (got.cpu() - same).abs().max().item()
    e_f32 = (got - f32).abs().max().item()
    print(f"fp16 window attention {B, H, W, ws, shift, nH}: vs same-rounding oracle {e_same:.2e}, vs fp32 operator {e_f32:.2e}")
    assert e_same < WINDOW_F16_ATOL_SAME_ROUNDING
    assert 0 < e_f32 < WINDOW_F16_ATOL
    with pytest.raises(ValueError):
        ops.window_attention_image(*dev, *args, mma="bf16")


@pytest.mark.parametrize("N,Lq,M,shapes,bcast", [(2, 37, 8, [(8, 14), (16, 28), (32, 56)], True), (1, 5, 2, [(3, 4)], False),
                                                  (3, 11, 4, [(2, 2), (4, 4), (8, 8), (16, 16)], False)], ids=lambda v: str(v))
def test_msda_prepare_matches_reference_expressions(cuda, N, Lq, M, shapes, bcast):
    """ops.msda_prepare == softmax + reference + offset / normalizer (ms_deform_attn.py:100-113), merged or padded
    projection rows, broadcast and per-batch reference points."""
    from oracle import cpu_path
    L, P = len(shapes), 4
    n_off = M * L * P * 2
    C = n_off + M * L * P + (8 if not bcast else 0)          # extra columns after the logits are ignored
    proj = synth.normal(f"prep/{N}/{Lq}/{M}/{L}", (N, Lq, C)) * 2.0
    ref = synth.uniform(f"prep/ref/{N}/{Lq}/{L}", (1 if bcast else N, Lq, L, 2), 0.0, 1.0)
    loc_r, attn_r = cpu_path.msda_prepare(proj, n_off, ref, shapes, M, L, P)
    loc, attn = ops.msda_prepare(proj.to(cuda), n_off, ref.to(cuda), shapes, M, L, P)
    assert (loc.cpu() - loc_r).abs().max().item() < 1e-6
    assert (attn.cpu() - attn_r).abs().max().item() < 1e-6
    assert loc.is_contiguous() and attn.is_contiguous()


@pytest.mark.parametrize("M,K,N,relu,bias", [(96600, 256, 256, False, True), (96600, 256, 288, False, True),
                                             (5000, 256, 1024, True, True), (4099, 128, 100, False, False),
                                             (2048, 384, 4, False, True), (3000, 256, 108, True, True)],
                         ids=lambda v: str(v))
@pytest.mark.parametrize("terms", [6, 3], ids=["bf16x6", "f16x3"])
def test_linear_split_matches_torch(cuda, linear_terms, terms, M, K, N, relu, bias):
    """ops.linear_split (fp32 from an exact 3-way bf16 split on the bf16 matrix cores: six products; or from two row-scaled
    fp16 parts: three products) == F.linear to fp32 rounding: within a few ulp, same order as the error of ATen's own fp32
    GEMM against the fp64 result."""
    linear_terms(terms)
    F = torch.nn.functional
    x = synth.normal(f"ls/x/{M}x{K}", (M, K), std=1.0)
    w = synth.normal(f"ls/w/{N}x{K}", (N, K), std=K ** -0.5)
    b = synth.normal(f"ls/b/{N}", (N,), std=0.5) if bias else None
    xd, wd, bd = x.to(cuda), w.to(cuda), (b.to(cuda) if bias else None)
    y = ops.linear_split(xd.view(4 if M % 4 == 0 else 1, -1, K), wd, bd, relu=relu)
    assert y is not None and tuple(y.shape) == (4 if M % 4 == 0 else 1, M // (4 if M % 4 == 0 else 1), N)
    ref64 = F.linear(xd.double(), wd.double(), bd.double() if bias else None)
    ref32 = F.linear(xd, wd, bd)
    if relu:
        ref64, ref32 = ref64.relu(), ref32.relu()
    err = (y.view(M, N).double() - ref64).abs().max().item()
    err32 = (ref32.double() - ref64).abs().max().item()
    assert err < max(4.0 * err32, 5e-6), (err, err32)        # a few ulp of the result, like any fp32 GEMM
    # scaling by a power of two is exact in the split as in fp32
    y2 = ops.linear_split(2.0 * xd, wd, None)
    y1 = ops.linear_split(xd, wd, None)
    assert torch.equal(y2, 2.0 * y1)


@pytest.mark.parametrize("M,K,N,act,res", [(58880, 96, 288, None, False), (58880, 96, 384, "gelu", False), (14720, 384, 96, None, True),
                                            (14720, 192, 576, None, False), (14720, 192, 768, "gelu", False), (3680, 768, 192, None, True),
                                            (3680, 384, 1536, "gelu", False), (4099, 96, 100, "relu", False), (2500, 768, 2304, None, False),
                                            (19320, 1024, 256, None, False), (4600, 1536, 384, None, True), (4613, 3072, 768, "gelu", False),
                                            (5000, 1024, 128, "relu", False)],
                         ids=lambda v: str(v))
@pytest.mark.parametrize("terms", [6, 3], ids=["bf16x6", "f16x3"])
def test_linear_fused_matches_torch(cuda, linear_terms, terms, M, K, N, act, res):
    """ops.linear_fused at the Swin-T widths (K = 96 / 192: register ring of three k-steps; 384 / 768: ring of four) with
    the fused epilogues (exact GELU, residual add) == F.linear + F.gelu / + residual to fp32 rounding."""
    linear_terms(terms)
    F = torch.nn.functional
    x = synth.normal(f"lf/x/{M}x{K}", (M, K), std=1.0)
    w = synth.normal(f"lf/w/{N}x{K}", (N, K), std=K ** -0.5)
    b = synth.normal(f"lf/b/{N}", (N,), std=0.5)
    r = synth.normal(f"lf/r/{M}x{N}", (M, N), std=1.0) if res else None
    xd, wd, bd, rd = x.to(cuda), w.to(cuda), b.to(cuda), (r.to(cuda) if res else None)
    y = ops.linear_fused(xd, wd, bd, act=act, residual=rd)
    assert y is not None and tuple(y.shape) == (M, N)
    ref64 = F.linear(xd.double(), wd.double(), bd.double())
    ref32 = F.linear(xd, wd, bd)
    if act == "gelu":
        ref64, ref32 = F.gelu(ref64), F.gelu(ref32)
    if act == "relu":
        ref64, ref32 = ref64.relu(), ref32.relu()
    if res:
        ref64, ref32 = ref64 + rd.double(), ref32 + rd
    err = (y.double() - ref64).abs().max().item()
    err32 = (ref32.double() - ref64).abs().max().item()
    assert err < max(4.0 * err32, 5e-6), (err, err32)
    with pytest.raises(RuntimeError):
        ops.linear_fused(xd, wd, bd, act="tanh")
    assert ops.linear_fused(xd, wd, bd, act="gelu", residual=torch.zeros(M, N, device=cuda)) is None   # not both


@pytest.mark.parametrize("M,C,Hd,act,res", [(19320, 256, 1024, "relu", False), (58880, 96, 384, "gelu", True), (14720, 192, 768, "gelu", True),
                                             (5000, 128, 512, "gelu", True), (4099, 96, 384, "gelu", False), (2049, 256, 32, "relu", True),
                                             (3000, 256, 2048, "gelu", False), (2048, 192, 96, "relu", False), (18400, 384, 1536, "gelu", True),
                                             (2100, 384, 64, "relu", False)], ids=lambda v: str(v))
def test_mlp_fused_matches_torch(cuda, M, C, Hd, act, res):
    """ops.mlp_fused (csrc/mlp_f16x3.hip: both Linears of an MLP in one kernel, hidden activations in registers, W2 pre-split
    in the k-order of the first product's accumulators) == linear -> activation -> linear (+ residual) to fp32 rounding:
    against fp64 the error is of the order of ATen's own fp32 path (msdeformattn.py:87-91, swin.py:35-58, :291-293)."""
    F = torch.nn.functional
    x = synth.normal(f"mlp/x/{M}x{C}", (M, C), std=1.0)
    w1 = synth.normal(f"mlp/w1/{Hd}x{C}", (Hd, C), std=C ** -0.5)
    b1 = synth.normal(f"mlp/b1/{Hd}", (Hd,), std=0.5)
    w2 = synth.normal(f"mlp/w2/{C}x{Hd}", (C, Hd), std=Hd ** -0.5)
    b2 = synth.normal(f"mlp/b2/{C}", (C,), std=0.5)
    r = synth.normal(f"mlp/r/{M}x{C}", (M, C), std=1.0) if res else None
    xd, w1d, b1d, w2d, b2d = (t.to(cuda) for t in (x, w1, b1, w2, b2))
    rd = r.to(cuda) if res else None
    y = ops.mlp_fused(xd, w1d, b1d, w2d, b2d, act, residual=rd)
    assert y is not None and tuple(y.shape) == (M, C)
    fn = F.relu if act == "relu" else F.gelu
    ref64 = F.linear(fn(F.linear(xd.double(), w1d.double(), b1d.double())), w2d.double(), b2d.double())
    ref32 = F.linear(fn(F.linear(xd, w1d, b1d)), w2d, b2d)
    if res:
        ref64, ref32 = ref64 + rd.double(), ref32 + rd
    err = (y.double() - ref64).abs().max().item()
    err32 = (ref32.double() - ref64).abs().max().item()
    assert err < max(4.0 * err32, 5e-6), (err, err32)
    # the two-kernel path it replaces computes the same thing with the same arithmetic: equal to rounding
    two = ops.linear_fused(ops.linear_fused(xd, w1d, b1d, act=act), w2d, b2d, residual=rd)
    if two is not None:
        assert (two - y).abs().max().item() < max(4.0 * err32, 5e-6)
    # leading dimensions are kept; no biases; scaling x by a power of two under ReLU is exact
    y3 = ops.mlp_fused(xd.view(1, M, C), w1d, None, w2d, None, act)
    assert tuple(y3.shape) == (1, M, C)
    if act == "relu":
        assert torch.equal(ops.mlp_fused(4.0 * xd, w1d, None, w2d, None, act), 4.0 * y3.view(M, C))


@pytest.mark.parametrize("M,C,Hd", [(58880, 96, 384), (14720, 192, 768), (4099, 256, 512), (3000, 128, 512), (3680, 384, 1536)],
                         ids=lambda v: str(v))
def test_mlp_fused_with_layer_norm(cuda, M, C, Hd):
    """ops.mlp_fused(..., ln=...) == x + fc2(gelu(fc1(LayerNorm(x)))): the Swin block's norm2 + Mlp + shortcut (swin.py:289-293)
    as one launch, the LayerNorm evaluated on the x tile in registers.  Rows with offsets far from zero (mean >> std) included."""
    F = torch.nn.functional
    x = synth.normal(f"mlpln/x/{M}x{C}", (M, C), std=1.0)
    x[::7] += 30.0
    x[3::11] *= 40.0
    g_ = 1.0 + 0.2 * synth.normal(f"mlpln/g/{C}", (C,))
    b_ = 0.1 * synth.normal(f"mlpln/b/{C}", (C,))
    w1 = synth.normal(f"mlpln/w1/{Hd}x{C}", (Hd, C), std=C ** -0.5)
    b1 = synth.normal(f"mlpln/b1/{Hd}", (Hd,), std=0.5)
    w2 = synth.normal(f"mlpln/w2/{C}x{Hd}", (C, Hd), std=Hd ** -0.5)
    b2 = synth.normal(f"mlpln/b2/{C}", (C,), std=0.5)
    xd, gd, bd, w1d, b1d, w2d, b2d = (t.to(cuda) for t in (x, g_, b_, w1, b1, w2, b2))
    y = ops.mlp_fused(xd, w1d, b1d, w2d, b2d, "gelu", residual=xd, ln=(gd, bd, 1e-5))
    assert y is not None and tuple(y.shape) == (M, C)
    h64 = F.layer_norm(xd.double(), (C,), gd.double(), bd.double(), 1e-5)
    ref64 = xd.double() + F.linear(F.gelu(F.linear(h64, w1d.double(), b1d.double())), w2d.double(), b2d.double())
    ref32 = xd + F.linear(F.gelu(F.linear(F.layer_norm(xd, (C,), gd, bd, 1e-5), w1d, b1d)), w2d, b2d)
    two = xd + ops.mlp_fused(ops.layer_norm(xd, gd, bd, 1e-5), w1d, b1d, w2d, b2d, "gelu")
    err = (y.double() - ref64).abs().max().item()
    err32 = (ref32.double() - ref64).abs().max().item()
    err2 = (two.double() - ref64).abs().max().item()
    print(f"mlp_fused + LN {M, C, Hd}: {err:.2e} (ATen fp32 {err32:.2e}, LN kernel + fused MLP {err2:.2e})")
    assert err < max(4.0 * err32, 2e-5), (err, err32)
    if C <= 256:
        # dual output: the block's result and the NEXT LayerNorm of it (the next block's norm1 / the stage's output norm) from one launch
        g2 = (1.0 + 0.2 * synth.normal(f"mlpln/g2/{C}", (C,))).to(cuda)
        b2n = (0.1 * synth.normal(f"mlpln/b2n/{C}", (C,))).to(cuda)
        pair = ops.mlp_fused(xd, w1d, b1d, w2d, b2d, "gelu", residual=xd, ln=(gd, bd, 1e-5), post_ln=(g2, b2n, 1e-5), dual=True)
        assert pair is not None and torch.equal(pair[0], y)
        n64 = F.layer_norm(y.double(), (C,), g2.double(), b2n.double(), 1e-5)
        e_n = (pair[1].double() - n64).abs().max().item()
        e_k = (ops.layer_norm(y, g2, b2n, 1e-5).double() - n64).abs().max().item()
        assert e_n < max(4.0 * e_k, 2e-5), (e_n, e_k)
        with pytest.raises(RuntimeError):
            ops.mlp_fused(xd, w1d, b1d, w2d, b2d, "gelu", residual=xd, dual=True)


@pytest.mark.parametrize("N,S,C,Hd", [(2, 9660, 256, 1024), (3, 1111, 256, 512), (1, 4000, 192, 384)], ids=lambda v: str(v))
def test_mlp_fused_post_norm(cuda, N, S, C, Hd):
    """ops.mlp_fused(..., residual=x, post_ln=..., post_add=pos) == (y, y + pos) with y = LayerNorm(x + linear2(relu(linear1(x)))):
    the tail of the MSDeformAttn encoder layer (msdeformattn.py:87-95) and the next layer's `with_pos_embed` as one launch; pos
    [1, S, C] broadcast over the N frames; also without post_add."""
    F = torch.nn.functional
    x = synth.normal(f"mlppn/x/{N}x{S}x{C}", (N, S, C))
    x[:, ::5] += 3.0
    pos = synth.normal(f"mlppn/pos/{S}x{C}", (1, S, C))
    g_ = 1.0 + 0.2 * synth.normal(f"mlppn/g/{C}", (C,))
    b_ = 0.1 * synth.normal(f"mlppn/b/{C}", (C,))
    w1 = synth.normal(f"mlppn/w1/{Hd}x{C}", (Hd, C), std=C ** -0.5)
    b1 = synth.normal(f"mlppn/b1/{Hd}", (Hd,), std=0.5)
    w2 = synth.normal(f"mlppn/w2/{C}x{Hd}", (C, Hd), std=Hd ** -0.5)
    b2 = synth.normal(f"mlppn/b2/{C}", (C,), std=0.5)
    xd, pd, gd, bd, w1d, b1d, w2d, b2d = (t.to(cuda) for t in (x, pos, g_, b_, w1, b1, w2, b2))
    res = ops.mlp_fused(xd, w1d, b1d, w2d, b2d, "relu", residual=xd, post_ln=(gd, bd, 1e-5), post_add=pd)
    assert res is not None and len(res) == 2
    y, y2 = res
    ffn64 = F.linear(F.relu(F.linear(xd.double(), w1d.double(), b1d.double())), w2d.double(), b2d.double())
    ref64 = F.layer_norm(xd.double() + ffn64, (C,), gd.double(), bd.double(), 1e-5)
    ref32 = F.layer_norm(xd + F.linear(F.relu(F.linear(xd, w1d, b1d)), w2d, b2d), (C,), gd, bd, 1e-5)
    err = (y.double() - ref64).abs().max().item()
    err32 = (ref32.double() - ref64).abs().max().item()
    print(f"mlp_fused post-norm {N, S, C, Hd}: {err:.2e} (ATen fp32 {err32:.2e})")
    assert tuple(y.shape) == (N, S, C) and err < max(4.0 * err32, 5e-6), (err, err32)
    assert torch.equal(y2, y + pd)
    y_only = ops.mlp_fused(xd, w1d, b1d, w2d, b2d, "relu", residual=xd, post_ln=(gd, bd, 1e-5))
    assert torch.equal(y_only, y)
    with pytest.raises(RuntimeError):
        ops.mlp_fused(xd, w1d, b1d, w2d, b2d, "relu", post_add=pd)
    # the whole tail of the encoder layer behind `src + output_proj(...)`: x1 = norm1(x), y = norm2(x1 + ffn(x1)) -- norm1 on the x tile
    # in registers, its result also the residual (parked in y, read back for the epilogue); == the LayerNorm launch + the call above
    g1 = (1.0 + 0.2 * synth.normal(f"mlppn/g1/{C}", (C,))).to(cuda)
    b1n = (0.1 * synth.normal(f"mlppn/b1n/{C}", (C,))).to(cuda)
    res2 = ops.mlp_fused(xd, w1d, b1d, w2d, b2d, "relu", ln=(g1, b1n, 1e-5), residual_normed=True, post_ln=(gd, bd, 1e-5), post_add=pd)
    assert res2 is not None and len(res2) == 2
    x1 = ops.layer_norm(xd, g1, b1n, 1e-5)
    sep = ops.mlp_fused(x1, w1d, b1d, w2d, b2d, "relu", residual=x1, post_ln=(gd, bd, 1e-5))
    x1_64 = F.layer_norm(xd.double(), (C,), g1.double(), b1n.double(), 1e-5)
    ref2 = F.layer_norm(x1_64 + F.linear(F.relu(F.linear(x1_64, w1d.double(), b1d.double())), w2d.double(), b2d.double()), (C,), gd.double(),
                        bd.double(), 1e-5)
    x1_32 = F.layer_norm(xd, (C,), g1, b1n, 1e-5)
    ref2_32 = F.layer_norm(x1_32 + F.linear(F.relu(F.linear(x1_32, w1d, b1d)), w2d, b2d), (C,), gd, bd, 1e-5)
    e2, e2_32, e2_sep = ((t.double() - ref2).abs().max().item() for t in (res2[0], ref2_32, sep))
    print(f"mlp_fused norm1 + ffn + norm2 {N, S, C, Hd}: {e2:.2e} (ATen fp32 {e2_32:.2e}, LayerNorm launch + fused MLP {e2_sep:.2e})")
    assert e2 < max(4.0 * e2_32, 5e-6), (e2, e2_32)
    assert torch.equal(res2[1], res2[0] + pd) and torch.equal(xd.cpu(), x)
    with pytest.raises(RuntimeError):
        ops.mlp_fused(xd, w1d, b1d, w2d, b2d, "relu", residual=xd, ln=(g1, b1n, 1e-5), residual_normed=True)


@pytest.mark.parametrize("B,H,W,C", [(5, 184, 320, 96), (2, 92, 160, 192), (2, 46, 80, 384), (1, 23, 41, 128), (3, 7, 9, 768), (1, 1, 1, 4)],
                         ids=lambda v: str(v))
def test_patch_merge_norm(cuda, B, H, W, C):
    """ops.patch_merge_norm == PatchMerging.forward up to its Linear (swin.py:341-386): zero padding to even sizes, the four strided
    slices concatenated in the reference's order, LayerNorm over 4 C -- bit-identical to the LayerNorm kernel on the concatenated
    tensor (same lanes, same summation order), and to ATen's layer_norm within fp32 rounding."""
    F = torch.nn.functional
    x = synth.normal(f"pm/x/{B}x{H}x{W}x{C}", (B, H, W, C))
    x[:, ::3] += 2.0
    g_ = 1.0 + 0.2 * synth.normal(f"pm/g/{C}", (4 * C,))
    b_ = 0.1 * synth.normal(f"pm/b/{C}", (4 * C,))
    xd, gd, bd = x.to(cuda), g_.to(cuda), b_.to(cuda)
    y = ops.patch_merge_norm(xd, gd, bd, 1e-5)
    xp = F.pad(xd, (0, 0, 0, W % 2, 0, H % 2))
    cat = torch.cat([xp[:, 0::2, 0::2, :], xp[:, 1::2, 0::2, :], xp[:, 0::2, 1::2, :], xp[:, 1::2, 1::2, :]], -1).reshape(B, -1, 4 * C)
    assert y is not None and tuple(y.shape) == tuple(cat.shape)
    ref64 = F.layer_norm(cat.double(), (4 * C,), gd.double(), bd.double(), 1e-5)
    ref32 = F.layer_norm(cat, (4 * C,), gd, bd, 1e-5)
    err, err32 = (y.double() - ref64).abs().max().item(), (ref32.double() - ref64).abs().max().item()
    assert err < max(2.0 * err32, 2e-6), (err, err32)
    assert torch.equal(y, ops.layer_norm(cat.contiguous(), gd, bd, 1e-5))
    assert ops.patch_merge_norm(torch.zeros(1, 4, 4, 6, device=cuda), torch.ones(24, device=cuda), torch.zeros(24, device=cuda)) is None


@pytest.mark.parametrize("T,Q,h,w", [(5, 100, 46, 80), (2, 130, 23, 40), (3, 17, 92, 160)], ids=lambda v: str(v))
def test_deferred_attention_mask(cuda, T, Q, h, w):
    """mask_decode_attn(deferred=True): the contraction without the flag memset and the row-reset pass; `materialize()` == the eager
    mask bit for bit; cross_attention on the DeferredMask == cross_attention on the eager mask bit for bit -- with rows whose
    every key is masked (the ...decoder_univs.py:390 rule: such a row attends to all keys), on a flags buffer that is reused by
    consecutive calls (stale generations must not leak)."""
    C, H = 256, 8
    feat = synth.normal(f"dm/f/{T}x{h}x{w}", (T, C, h, w))
    feat[:, 0] = feat[:, 0].abs() + 1.0                          # a strictly positive channel
    feat = feat.to(cuda)
    S = h * w
    k = synth.normal(f"dm/k/{T}x{S}", (S, T, 256)).to(cuda)
    v = synth.normal(f"dm/v/{T}x{S}", (S, T, 256)).to(cuda)
    q = synth.normal(f"dm/q/{T}x{Q}", (Q, T, 256)).to(cuda)
    for rnd in range(3):
        me = synth.normal(f"dm/me/{T}x{Q}/{rnd}", (T, Q, C))
        # rows that end up fully masked: logits negative everywhere (different rows every round)
        for r in range(rnd, Q, 7):
            me[:, r] = 0.0
            me[:, r, 0] = -100.0
        med = me.to(cuda)
        eager = ops.mask_decode_attn(med, feat)
        dm = ops.mask_decode_attn(med, feat, deferred=True)
        assert isinstance(dm, ops.DeferredMask) and tuple(dm.shape) == (T, Q, S)
        raw_full = (dm.mask.view(T * Q, S) != 0).all(dim=1)
        assert int(raw_full.sum()) >= T * len(range(rnd, Q, 7)) and not eager.view(T * Q, S)[raw_full].any()
        if S >= 512:
            a = ops.cross_attention(q, k, v, dm, H, 32 ** -0.5)
            b = ops.cross_attention(q, k, v, eager, H, 32 ** -0.5)
            assert a is not None and torch.equal(a, b)
        assert torch.equal(dm.materialize(), eager)
    # held masks stay valid however many deferred masks of the same kind follow them (ring of flag buffers; a buffer's last mask is
    # made explicit before the buffer serves a new generation): two decoder instances interleaving on one stream
    held = [ops.mask_decode_attn(med, feat, deferred=True) for _ in range(3)]
    others = []
    for rnd in range(7):                                          # more than a ring's worth of newer generations, all kept alive
        me2 = synth.normal(f"dm/me2/{T}x{Q}/{rnd}", (T, Q, C)).to(cuda)
        others.append((ops.mask_decode_attn(me2, feat, deferred=True), ops.mask_decode_attn(me2, feat)))
    for dm in held:
        if S >= 512:
            assert torch.equal(ops.cross_attention(q, k, v, dm, H, 32 ** -0.5), ops.cross_attention(q, k, v, eager, H, 32 ** -0.5))
        assert torch.equal(dm.materialize(), eager)
    for dm, eg in others:
        assert torch.equal(dm.materialize(), eg)


def test_tokens_from_nchw(cuda):
    """ops.tokens_from_nchw: the levels' NCHW maps -> the encoder input [T, S, C] in one launch per level, GroupNorm applied on the way
    (== group_norm + transpose + concatenation bit for bit), `src + pos` as a second output (== the add bit for bit); levels without a
    GroupNorm; shapes that are not covered (msdeformattn.py:168-188, :205-212, :61-63)."""
    T, C = 3, 256
    shapes = [(6, 10), (12, 20), (23, 40)]
    xs = [synth.normal(f"tok/x/{h}x{w}", (T, C, h, w)).to(cuda) * 3.0 + 0.5 for h, w in shapes]
    S = sum(h * w for h, w in shapes)
    pos = synth.normal("tok/pos", (1, S, C)).to(cuda)
    g_ = (1.0 + 0.2 * synth.normal("tok/g", (C,))).to(cuda)
    b_ = (0.1 * synth.normal("tok/b", (C,))).to(cuda)
    affs = [ops.group_norm_affine(x, 32, g_, b_, 1e-5) for x in xs]
    out = ops.tokens_from_nchw(xs, affs, pos)
    assert out is not None
    src, q0 = out
    ref = torch.cat([ops.transpose_last2(ops.group_norm(x, 32, g_, b_, 1e-5).flatten(2)) for x in xs], 1)
    assert tuple(src.shape) == (T, S, C) and torch.equal(src, ref) and torch.equal(q0, ref + pos)
    src2, none = ops.tokens_from_nchw(xs, [None, affs[1], None], None)
    assert none is None
    ref2 = torch.cat([ops.transpose_last2((x if i != 1 else ops.group_norm(x, 32, g_, b_, 1e-5)).flatten(2)) for i, x in enumerate(xs)], 1)
    assert torch.equal(src2, ref2)
    assert ops.tokens_from_nchw([xs[0][..., :9]], [None], None) is None          # H * W % 4 != 0


@pytest.mark.parametrize("M,K,Nw,rows,add,relu,res,ln", [
    (500, 256, 768, (0, 256), True, False, False, False),      # cross-attention q = in_proj_q(tgt + query_pos)
    (500, 256, 768, (0, 512), True, False, False, False),      # self-attention q, k in one projection
    (500, 256, 768, (512, 256), False, False, False, False),   # self-attention v
    (500, 256, 256, None, False, False, True, True),           # norm(tgt + out_proj(attn))
    (550, 256, 2048, None, False, True, False, False),         # relu(linear1(tgt))
    (300, 128, 256, None, True, True, True, True),
    (2000, 256, 256, None, False, True, False, False),         # mask-embedding MLP at config 5's length
    (7, 64, 48, (16, 32), True, True, True, False), (1, 32, 16, None, False, False, False, False), (4096, 256, 256, None, True, False, True, True),
    (500, 256, 768, (0, 768), "qk", False, False, False),     # q, k (with the position embedding) and v (without) in one launch
], ids=lambda v: str(v))
def test_small_linear_matches_torch(cuda, M, K, Nw, rows, add, relu, res, ln):
    """ops.small_linear (csrc/small_linear.hip: a few-rows Linear with `x + x_add` in front and ReLU / residual / LayerNorm behind, rows of
    a packed weight selected by offset) == the torch expression in fp64 to fp32 rounding (transformer_layers.py:30-46, :95-115, :150-166)."""
    F = torch.nn.functional
    tag = f"sl/{M}x{K}x{Nw}"
    x = synth.normal(tag + "/x", (M, K))
    x[::3] *= 30.0
    xa = synth.normal(tag + "/xa", (M, K))
    w = synth.normal(tag + "/w", (Nw, K), std=K ** -0.5)
    b = synth.normal(tag + "/b", (Nw,), std=0.5)
    f0, N = (0, Nw) if rows is None else rows
    r = synth.normal(tag + "/r", (M, N))
    g_ = 1.0 + 0.2 * synth.normal(tag + "/g", (N,))
    be = 0.1 * synth.normal(tag + "/be", (N,))
    xd, xad, wd, bd, rd, gd, bed = (t.to(cuda) for t in (x, xa, w, b, r, g_, be))
    y = ops.small_linear(xd, wd, bd, rows=rows, x_add=xad if add else None, relu=relu, residual=rd if res else None,
                         ln=(gd, bed, 1e-5) if ln else None, add_features=512 if add == "qk" else 0)
    assert y is not None and tuple(y.shape) == (M, N)

    def ref(dt):
        c = lambda t: t.to(dt)
        if add == "qk":
            t = torch.cat([F.linear(c(xd) + c(xad), c(wd)[:512], c(bd)[:512]), F.linear(c(xd), c(wd)[512:], c(bd)[512:])], -1)
        else:
            t = F.linear(c(xd) + c(xad) if add else c(xd), c(wd)[f0:f0 + N], c(bd)[f0:f0 + N])
        if relu:
            t = F.relu(t)
        if res:
            t = t + c(rd)
        return F.layer_norm(t, (N,), c(gd), c(bed), 1e-5) if ln else t
    ref64, ref32 = ref(torch.float64), ref(torch.float32)
    scale = max(1.0, ref64.abs().max().item())
    err, err32 = (y.double() - ref64).abs().max().item() / scale, (ref32.double() - ref64).abs().max().item() / scale
    assert err < max(4.0 * err32, 2e-6), (err, err32)
    # leading dimensions are kept; no bias; shapes that are not covered
    y3 = ops.small_linear(xd.view(1, M, K), wd, None, rows=rows)
    assert tuple(y3.shape) == (1, M, N)
    if M % 5 == 0 and not res and not ln:
        # rows as (q, t) pairs handed back as [t, q]: the mask embeddings' layout change inside the kernel's store
        yt = ops.small_linear(xd.view(M // 5, 5, K), wd, bd, rows=rows, x_add=xad.view(M // 5, 5, K) if add is True else None, relu=relu,
                              transpose01=True)
        if add is True or not add:
            assert tuple(yt.shape) == (5, M // 5, N) and yt.is_contiguous() and torch.equal(yt, y.view(M // 5, 5, N).transpose(0, 1))
    assert ops.small_linear(torch.zeros(5000, K, device=cuda), wd, bd) is None           # too many rows (the tall kernels' job)
    assert ops.small_linear(xd[:, :K - 8].contiguous(), wd[:, :K - 8].contiguous(), bd) is None or (K - 8) % 32 == 0


def test_small_linear_module_paths(cuda):
    """layers.MultiheadAttention / FFNLayer / MLP through the few-rows kernel == the library GEMM + elementwise launches they replace
    (SWITCHES.small_linear off) to fp32 rounding: self-attention with positional queries and a mask, cross-attention with precomputed
    key / value projections, the post-norm FFN, the three-layer mask-embedding MLP."""
    from univs_amd import layers
    from univs_amd.modeling.transformer_decoder import transformer_layers as tl
    from univs_amd.switches import override
    E, Hh, L, S = 256, 8, 500, 920
    sa = tl.SelfAttentionLayer(E, Hh).to(cuda).eval()
    ca = tl.CrossAttentionLayer(E, Hh).to(cuda).eval()
    ffn = tl.FFNLayer(E, 2048).to(cuda).eval()
    mlp = layers.MLP(E, E, E, 3).to(cuda).eval()
    with torch.no_grad():
        for mod, nm in ((sa, "sa"), (ca, "ca"), (ffn, "ffn"), (mlp, "mlp")):
            for n_, p_ in mod.named_parameters():
                p_.copy_((synth.normal(f"slm/{nm}/{n_}", tuple(p_.shape), std=0.06) + (1.0 if n_.endswith("norm.weight") else 0.0)).to(cuda))
        tgt = synth.normal("slm/tgt", (L, 1, E)).to(cuda)
        pos = synth.normal("slm/pos", (L, 1, E)).to(cuda)
        mem = synth.normal("slm/mem", (S, 1, E)).to(cuda)
        msk = (torch.rand(L, L, generator=torch.Generator().manual_seed(2)) < 0.3).to(cuda)
        msk[torch.arange(L), torch.arange(L)] = False
        cm = (torch.rand(1, L, S, generator=torch.Generator().manual_seed(4)) < 0.5).to(cuda)

        def run():
            a = sa(tgt, tgt_mask=msk, query_pos=pos)
            b = ca(a, mem, memory_mask=cm, query_pos=pos)
            c = ffn(b)
            return a, b, c, mlp(c)
        small = run()
        with override(small_linear=False):
            plain = run()
    for s_, p_, nm in zip(small, plain, ("self-attention", "cross-attention", "ffn", "mlp")):
        assert (s_ - p_).abs().max().item() < 3e-5, (nm, (s_ - p_).abs().max().item())


def test_mlp_fused_row_scaling_and_uncovered_shapes(cuda):
    """Rows of x over 80 binades, zero rows, hidden rows whose magnitude jumps between chunks (the running scale of the hidden
    activations is lowered with an exact rescaling of the output accumulators), Inf / NaN confined to their row; shapes the
    kernel does not cover return None."""
    F = torch.nn.functional
    M, C, Hd = 4096, 256, 512
    g = torch.Generator().manual_seed(11)
    x = torch.randn(M, C, generator=g) * torch.exp2(torch.randint(-30, 30, (M, 1), generator=g).float())
    x[5] = 0
    x[8, 3] = 1e4 * x[8].abs().max()
    w1 = torch.randn(Hd, C, generator=g) / 16
    w1[:32] *= 2.0 ** -20                 # first chunk tiny, later chunks large: the hidden scale must be lowered on the way
    w1[320:352] *= 2.0 ** 12
    b1 = torch.randn(Hd, generator=g) * 0.1
    w2 = torch.randn(C, Hd, generator=g) * torch.exp2(torch.randint(-6, 6, (C, 1), generator=g).float()) / 22
    b2 = torch.randn(C, generator=g)
    xd, w1d, b1d, w2d, b2d = (t.to(cuda) for t in (x, w1, b1, w2, b2))
    for act, fn in (("relu", F.relu), ("gelu", F.gelu)):
        y = ops.mlp_fused(xd, w1d, b1d, w2d, b2d, act)
        assert y is not None and torch.isfinite(y).all()
        h64 = fn(F.linear(xd.double(), w1d.double(), b1d.double()))
        ref64 = F.linear(h64, w2d.double(), b2d.double())
        ref32 = F.linear(fn(F.linear(xd, w1d, b1d)), w2d, b2d)
        scale = h64.abs() @ w2d.double().abs().t() + b2d.double().abs()[None] + 1e-300
        e3 = ((y.double() - ref64).abs() / scale).max().item()
        e32 = ((ref32.double() - ref64).abs() / scale).max().item()
        print(f"mlp_fused {act}: max error / sum|h||w2| {e3:.2e} (ATen fp32: {e32:.2e})")
        assert e3 < max(3.0 * e32, 5e-7), (act, e3, e32)
        xd2 = xd.clone()
        xd2[100, 17] = float("inf")
        xd2[200, 5] = float("nan")
        y2 = ops.mlp_fused(xd2, w1d, b1d, w2d, b2d, act)
        bad = torch.zeros(M, dtype=torch.bool, device=cuda)
        bad[100] = bad[200] = True
        assert torch.equal(y2[~bad], y[~bad]) and not torch.isfinite(y2[100]).any() and torch.isnan(y2[200]).all()
    assert ops.mlp_fused(torch.zeros(4096, 768, device=cuda), torch.zeros(3072, 768, device=cuda), None,
                         torch.zeros(768, 3072, device=cuda), None, "gelu") is None                              # C = 768
    assert ops.mlp_fused(xd[:100], w1d, b1d, w2d, b2d, "relu") is None                                           # few rows
    assert ops.mlp_fused(xd, w1d[:500], b1d[:500], w2d[:, :500].contiguous(), b2d, "relu") is None               # Hd % 32
    assert ops.mlp_fused(xd, w1d, b1d, w2d, b2d, "tanh") is None
    assert ops.mlp_fused(xd.cpu(), w1d.cpu(), None, w2d.cpu(), None, "relu") is None


def test_linear_f16x3_row_scaling(cuda, linear_terms):
    """The three-product Linear (linear_f16x3.hip) scales every row of x and of W into fp16's range by a power of two: rows
    of wildly different magnitude, rows with outliers, zero rows and tiny / huge rows keep the accuracy of the fp32 GEMM
    relative to THEIR OWN sum of |x||w| (the quantity a GEMM's rounding error scales with); Inf / NaN stay in their row."""
    linear_terms(3)
    M, K, N = 4096, 256, 288
    g = torch.Generator().manual_seed(7)
    x = torch.randn(M, K, generator=g)
    x *= torch.exp2(torch.randint(-40, 40, (M, 1), generator=g).float())          # rows over 80 binades
    x[5] = 0
    x[6] = torch.randn(K, generator=g) * 1e-30
    x[7] = torch.randn(K, generator=g) * 1e30
    x[8, 3] = 1e4 * x[8].abs().max()                                                # an outlier 10^4 above its row
    w = torch.randn(N, K, generator=g) * torch.exp2(torch.randint(-12, 12, (N, 1), generator=g).float()) / 16
    w[9] = 0
    b = torch.randn(N, generator=g)
    xd, wd, bd = x.to(cuda), w.to(cuda), b.to(cuda)
    y = ops.linear_split(xd, wd, bd)
    assert y is not None
    ref64 = torch.nn.functional.linear(xd.double(), wd.double(), bd.double())
    ref32 = torch.nn.functional.linear(xd, wd, bd)
    scale = xd.double().abs() @ wd.double().abs().t() + bd.double().abs()[None] + 1e-300
    e3 = ((y.double() - ref64).abs() / scale).max().item()
    e32 = ((ref32.double() - ref64).abs() / scale).max().item()
    print(f"f16x3 row scaling: max error / sum|x||w| {e3:.2e} (ATen fp32 GEMM: {e32:.2e})")
    assert e3 < max(2.0 * e32, 3e-7), (e3, e32)
    assert torch.equal(y[5], bd.expand(1, -1)[0]) and torch.isfinite(y).all()
    xd2 = xd.clone()
    xd2[100, 17] = float("inf")
    xd2[200, 5] = float("nan")
    y2 = ops.linear_split(xd2, wd, bd)
    bad = torch.zeros(M, dtype=torch.bool, device=cuda)
    bad[100] = bad[200] = True
    assert torch.equal(y2[~bad], y[~bad]) and not torch.isfinite(y2[100]).any() and torch.isnan(y2[200]).all()


@pytest.mark.parametrize("M,K,N,act,res,blocked", [(96600, 256, 256, None, False, 16), (96600, 256, 288, None, False, 36), (96600, 256, 256, None, True, 0),
                                                    (18400, 384, 1152, None, False, 0), (18400, 384, 1536, "gelu", False, 0), (58880, 96, 288, None, False, 0),
                                                    (14720, 192, 576, None, False, 0), (4099, 96, 100, "relu", False, 0), (2048, 384, 4, None, False, 0),
                                                    (73600, 256, 768, None, False, 0), (4600, 768, 2304, None, False, 0)], ids=lambda v: str(v))
def test_linear_resident_presplit_is_bit_identical(cuda, M, K, N, act, res, blocked):
    """The W-resident Linear staging its slab from the split image cached per weight tensor (univs_linear_resident_presplit_f32 /
    univs_linear_blocked_presplit_f32: a copy) == the same kernel splitting the slab inside every workgroup (univs_linear_fused_f32 /
    univs_linear_blocked_f32), bit for bit: same row maxima, same scales, same parts.  A weight VIEW keeps the raw-weight entry."""
    from univs_amd.switches import override
    x = synth.normal(f"rp/x/{M}x{K}", (M, K)).to(cuda)
    w = (synth.normal(f"rp/w/{N}x{K}", (N, K), std=K ** -0.5) * torch.logspace(-3, 3, N).view(N, 1)).to(cuda)   # rows over 20 binades
    b = synth.normal(f"rp/b/{N}", (N,)).to(cuda)
    r = synth.normal(f"rp/r/{M}x{N}", (M, N)).to(cuda) if res else None

    def run(wt):
        if blocked:
            return ops.linear_blocked(x.view(5, M // 5, K), wt, b, M // 5, blocked)
        return ops.linear_fused(x, wt, b, act=act, residual=r)
    with override(resident_presplit=False, presplit_kmin=0):
        want = run(w)
    with override(resident_presplit=True, presplit_kmin=0):
        got = run(w)
        wide = torch.cat([w, w], 0)
        view = run(wide[:N])                                  # a view: not cached, raw-weight entry
    assert want is not None and got is not None and torch.equal(got, want) and torch.equal(view, want)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    if act == "gelu":
        ref = torch.nn.functional.gelu(ref)
    if act == "relu":
        ref = ref.relu()
    if res:
        ref = ref + r.double()
    if blocked:
        ref = ref.view(5, M // 5, N // blocked, blocked).permute(0, 2, 1, 3)
    scale = (x.double().abs() @ w.double().abs().t()).max().item()
    assert (got.double() - ref.reshape(got.shape)).abs().max().item() < 2e-6 * scale


@pytest.mark.parametrize("M,A,stages", [(500, 100, 3), (17, 0, 3), (4000, 200, 2), (16, 0, 1), (1100, 0, 3)], ids=lambda v: str(v))
def test_small_mlp_chain_is_bit_identical_to_separate_launches(cuda, M, A, stages):
    """ops.small_mlp (csrc/small_linear.hip: small_chain_kernel -- up to three 256 -> 256 Linears of the mask-embedding MLP in ONE launch,
    the hidden rows handed over through LDS) == the same chain of ops.small_linear launches BIT FOR BIT (same row maxima, scales, parts),
    with the [A, B, C] -> [B, A, C] output order of the mask embeddings; with the `decoder_norm` LayerNorm inside the launch: the MLP output
    and the normalised rows against F.layer_norm + the chain to fp32 rounding, deterministic run to run; rows over many binades."""
    x = synth.normal(f"sm/x/{M}", (M, 256)) * torch.logspace(-4, 4, M).view(M, 1)
    if A:
        x = x.view(A, M // A, 256)
    x = x.to(cuda)
    lay = []
    for i in range(stages):
        w = synth.normal(f"sm/w{i}", (256, 256), std=1 / 16).to(cuda)
        b = synth.normal(f"sm/b{i}", (256,)).to(cuda) if i != 1 else None
        lay.append((w, b, i < stages - 1))
    want = x
    for i, (w, b, relu) in enumerate(lay):
        want = ops.small_linear(want, w, b, relu=relu, transpose01=bool(A) and i == stages - 1)
        assert want is not None
    got = ops.small_mlp(x, lay, transpose01=bool(A))
    assert got is not None and got.shape == want.shape and torch.equal(got, want)
    # the LayerNorm in front, inside the launch
    g_ = (1.0 + 0.3 * synth.normal("sm/g", (256,))).to(cuda)
    b_ = (0.2 * synth.normal("sm/bb", (256,))).to(cuda)
    r = ops.small_mlp(x, lay, in_ln=(g_, b_, 1e-5), want_normed=True, transpose01=bool(A))
    assert r is not None
    y, xn = r
    ref_n = torch.nn.functional.layer_norm(x.double(), (256,), g_.double(), b_.double(), 1e-5)
    assert xn.shape == x.shape and (xn.double() - ref_n).abs().max().item() < 5e-6
    ref = ref_n
    for w, b, relu in lay:
        ref = torch.nn.functional.linear(ref, w.double(), b.double() if b is not None else None)
        ref = ref.relu() if relu else ref
    if A:
        ref = ref.transpose(0, 1)
    assert (y.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    y2 = ops.small_mlp(x, lay, in_ln=(g_, b_, 1e-5), transpose01=bool(A))
    assert torch.equal(y2, y)                                    # without the second output: the same rows; and run to run
    assert ops.small_mlp(x.cpu(), [(w.cpu(), None, False) for w, _, _ in lay]) is None
    assert ops.small_mlp(x[..., :128].contiguous(), lay) is None


def test_linear_split_uncovered_shapes_return_none(cuda):
    x = torch.zeros(4096, 80, device=cuda)
    assert ops.linear_split(x, torch.zeros(96, 80, device=cuda)) is None                      # K % 96 and K % 128
    assert ops.linear_split(torch.zeros(4096, 1024, device=cuda), torch.zeros(6, 1024, device=cuda)) is None   # wide K: N % 4 too
    assert ops.linear_split(torch.zeros(4096, 256, device=cuda), torch.zeros(6, 256, device=cuda)) is None     # N % 4
    assert ops.linear_split(torch.zeros(100, 256, device=cuda), torch.zeros(8, 256, device=cuda)) is None      # few rows
    assert ops.linear_split(torch.zeros(4096, 256), torch.zeros(8, 256)) is None                               # CPU tensors
    from univs_amd import layers
    y = layers.linear(x, torch.ones(96, 80, device=cuda), None)                                # falls through to ATen
    assert tuple(y.shape) == (4096, 96)


@pytest.mark.parametrize("T,H,W,div", [(5, 720, 1280, 32), (2, 37, 53, 32), (1, 64, 96, 0), (3, 33, 64, 32)], ids=lambda v: str(v))
def test_normalize_pad_is_bit_identical(cuda, T, H, W, div):
    """ops.normalize_pad == ImageList.from_tensors([(f - mean) / std ...], size_divisibility) of the clip loop's pre-step
    (inference_video_entity.py:246-250) bit for bit: fp32 subtraction, true division, zero padding at the bottom / right; widths that
    are not multiples of four; the driver helper takes it for same-size GPU frames and the per-frame path otherwise."""
    from univs_amd.inference.video_entity import ImageList, normalized_image_list
    x = (synth.uniform(f"np/x/{T}x{H}x{W}", (T, 3, H, W), 0.0, 255.0)).to(cuda)
    mean = torch.tensor([123.675, 116.28, 103.53], device=cuda).view(3, 1, 1)
    std = torch.tensor([58.395, 57.12, 57.375], device=cuda).view(3, 1, 1)
    want = ImageList.from_tensors([(f - mean) / std for f in x], div).tensor
    got = ops.normalize_pad(x, mean, std, div)
    assert got is not None and got.shape == want.shape and torch.equal(got, want)
    il = normalized_image_list(list(x), mean, std, div)
    assert torch.equal(il.tensor, want) and il.image_sizes == [(H, W)] * T
    ragged = [x[0], x[1][:, : H - 1]] if T > 1 else [x[0]]
    il2 = normalized_image_list(ragged, mean, std, div)
    assert torch.equal(il2.tensor, ImageList.from_tensors([(f - mean) / std for f in ragged], div).tensor)
    assert ops.normalize_pad(x.cpu(), mean.cpu(), std.cpu(), div) is None


@pytest.mark.parametrize("T,H,W,E,norm", [(2, 736, 1280, 96, True), (1, 64, 96, 96, True), (3, 36, 52, 128, True), (1, 72, 40, 192, False),
                                          (2, 20, 44, 96, False)], ids=lambda v: str(v))
def test_patch_embed4_matches_torch(cuda, T, H, W, E, norm):
    """ops.patch_embed4 == LayerNorm(conv2d(x, w, b, stride 4).flatten(2).transpose(1, 2)) (swin.py:307-339) to fp32 rounding
    (plain fp32 FMAs, another summation order than the library's convolution)."""
    F = torch.nn.functional
    x = (synth.normal(f"pe4/x/{T}/{H}/{W}", (T, 3, H, W)) * 2.0).to(cuda)
    w = synth.normal(f"pe4/w/{E}", (E, 3, 4, 4), std=48 ** -0.5).to(cuda)
    b = synth.normal(f"pe4/b/{E}", (E,), std=0.3).to(cuda)
    g_ = (1.0 + 0.2 * synth.normal(f"pe4/g/{E}", (E,))).to(cuda)
    be = (0.1 * synth.normal(f"pe4/be/{E}", (E,))).to(cuda)
    got = ops.patch_embed4(x, w, b, (g_, be, 1e-5) if norm else None)
    assert got is not None and tuple(got.shape) == (T, (H // 4) * (W // 4), E)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=4).flatten(2).transpose(1, 2)
    if norm:
        ref = F.layer_norm(ref, (E,), g_.double(), be.double(), 1e-5)
    err = (got.double() - ref).abs().max().item()
    assert err < 2e-5, err
    assert ops.patch_embed4(x[:, :, : H - 1], w, b) is None                                   # H % 4
    from univs_amd.modeling.backbone.swin import PatchEmbed
    pe = PatchEmbed(4, 3, E, patch_norm=norm).to(cuda).eval()
    with torch.no_grad():
        pe.proj.weight.copy_(w)
        pe.proj.bias.copy_(b)
        if norm:
            pe.norm.weight.copy_(g_)
            pe.norm.bias.copy_(be)
        tok, Wh, Ww = pe.tokens(x[:, :, : H - 2, : W - 1])                                     # ragged image: padded as the module does
        nchw = pe(x[:, :, : H - 2, : W - 1])
    assert (Wh, Ww) == (nchw.size(2), nchw.size(3)) and (tok - nchw.flatten(2).transpose(1, 2)).abs().max().item() < 2e-5


@pytest.mark.parametrize("T,C,H,W", [(5, 256, 92, 160), (2, 256, 23, 40), (3, 64, 7, 12), (1, 256, 46, 80)], ids=lambda v: str(v))
def test_decoder_memory_is_exact(cuda, T, C, H, W):
    """ops.decoder_memory == the reference's expressions bit for bit: memory = (x.flatten(2) + level_embed[:, None]) permuted to
    [hw, t, C]; key = memory + pos with the 3-D sine embedding pos = yx + pos_t (...decoder_univs.py:350-355, :400-405); and the
    separable pieces of the embedding add up to the module's own forward."""
    from univs_amd.modeling.position_encoding import PositionEmbeddingSine3DArbitraryT
    x = synth.normal(f"dm/x/{T}/{C}/{H}/{W}", (T, C, H, W)).to(cuda)
    le = synth.normal(f"dm/le/{C}", (C,)).to(cuda)
    pe = PositionEmbeddingSine3DArbitraryT(C // 2, normalize=True)
    fi = torch.arange(3, 3 + T, device=cuda)[None]
    xi = x.view(1, T, C, H, W)
    yx, pz = pe.forward_separable(xi, fi)
    pos = pe(xi, fi)                                            # [1, T, C, H, W]
    assert torch.equal((yx.view(H, W, C)[None] + pz[0][:, None, None, :]).permute(0, 3, 1, 2), pos[0])
    got = ops.decoder_memory(x, le, yx, pz[0])
    assert got is not None
    mem, key = got
    s = x.flatten(2) + le[None, :, None]
    want_mem = s.permute(2, 0, 1).contiguous()
    want_key = (s.permute(2, 0, 1) + pos.flatten(3).flatten(0, 1).permute(2, 0, 1)).contiguous()
    assert torch.equal(mem, want_mem) and torch.equal(key, want_key)
    assert ops.decoder_memory(x[:, :, :, :W - 1].contiguous(), le, yx[:H * (W - 1)], pz[0]) is None or (W - 1) * H % 4 == 0


@pytest.mark.parametrize("shape", [(5, 58880, 96), (2, 920, 768), (3, 7, 96, 100), (1, 64, 64), (2, 10, 6), (1, 68, 132)], ids=str)
def test_transpose_last2_is_exact(cuda, shape):
    """ops.transpose_last2 == x.transpose(-2, -1).contiguous() bit for bit: LDS-tiled where both extents are multiples of
    4 (edge tiles included), ATen otherwise."""
    x = synth.normal(f"tr/{shape}", shape).to(cuda)
    got = ops.transpose_last2(x)
    assert got.is_contiguous() and torch.equal(got, x.transpose(-2, -1).contiguous())
    if x.dim() == 3 and x.shape[1] >= 12:
        # row ranges of the batch tensor (the per-level split of the pixel decoder's encoder output): read in place, no copy first
        for r0, r1 in ((0, 8), (4, x.shape[1]), (x.shape[1] - 4, x.shape[1])):
            z = x[:, r0:r1]
            assert not z.is_contiguous() or x.shape[0] == 1
            assert torch.equal(ops.transpose_last2(z), z.transpose(-2, -1).contiguous())


@pytest.mark.parametrize("T,Cin,Cout,H,W", [(2, 256, 256, 48, 44), (1, 128, 128, 70, 64), (3, 256, 256, 17, 83), (1, 384, 256, 64, 64)], ids=str)
def test_conv3x3_matches_torch(cuda, T, Cin, Cout, H, W):
    """ops.conv3x3 (3 x 3, stride 1, padding 1, no bias: the x-stationary GEMM with tap addressing, three fp16 products on
    weights split once per tensor) == F.conv2d to fp32 rounding, borders and ragged row tiles included."""
    F = torch.nn.functional
    x = synth.normal(f"cv/x/{T}/{Cin}/{H}/{W}", (T, Cin, H, W)).to(cuda)
    w = synth.normal(f"cv/w/{Cout}/{Cin}", (Cout, Cin, 3, 3), std=(9 * Cin) ** -0.5).to(cuda)
    y = ops.conv3x3(x, w)
    assert y is not None and tuple(y.shape) == (T, Cout, H, W)
    ref64 = F.conv2d(x.double(), w.double(), None, 1, 1)
    ref32 = F.conv2d(x, w, None, 1, 1)
    err = (y.double() - ref64).abs().max().item()
    err32 = (ref32.double() - ref64).abs().max().item()
    # (the library's own error depends on the solver MIOpen picks on the box -- 1.4e-6 ... 2.5e-6 at K = 2 304 -- so the floor, not its
    # multiple, is the bound that has to hold everywhere: 1e-5 = fp32 accumulation over 9 Cin terms of this magnitude)
    assert err < max(4.0 * err32, 1e-5), (err, err32)
    assert ops.conv3x3(torch.zeros(1, 96, 64, 64, device=cuda), torch.zeros(64, 96, 3, 3, device=cuda)) is None   # Cin % 128
    assert ops.conv3x3(torch.zeros(1, 128, 16, 16, device=cuda), torch.zeros(128, 128, 3, 3, device=cuda)) is None  # < 4096 pixels


@pytest.mark.parametrize("T,Cin,Cout,H,W,bias", [(2, 256, 256, 92, 160, True), (3, 96, 256, 60, 77, False), (1, 192, 256, 46, 93, True),
                                                  (2, 384, 256, 46, 80, True), (5, 768, 256, 23, 40, True), (1, 128, 64, 70, 70, False)],
                         ids=lambda v: str(v))
def test_conv1x1_matches_torch(cuda, T, Cin, Cout, H, W, bias):
    """ops.conv1x1 (the streamed three-product GEMM with the centre tap alone, bias in the epilogue) == F.conv2d to fp32 rounding:
    the lateral / mask-feature / input-projection convolutions of the pixel decoder (msdeformattn.py:205-232, :262-283)."""
    F = torch.nn.functional
    x = synth.normal(f"c1/x/{T}/{Cin}/{H}/{W}", (T, Cin, H, W)).to(cuda)
    w = synth.normal(f"c1/w/{Cout}/{Cin}", (Cout, Cin, 1, 1), std=Cin ** -0.5).to(cuda)
    b = synth.normal(f"c1/b/{Cout}", (Cout,)).to(cuda) if bias else None
    y = ops.conv1x1(x, w, b)
    assert y is not None and tuple(y.shape) == (T, Cout, H, W)
    ref64 = F.conv2d(x.double(), w.double(), b.double() if bias else None)
    ref32 = F.conv2d(x, w, b)
    err = (y.double() - ref64).abs().max().item()
    err32 = (ref32.double() - ref64).abs().max().item()
    assert err < max(4.0 * err32, 5e-6), (err, err32)
    assert ops.conv1x1(torch.zeros(1, 80, 64, 64, device=cuda), torch.zeros(64, 80, 1, 1, device=cuda)) is None      # Cin
    assert ops.conv1x1(torch.zeros(1, 128, 16, 16, device=cuda), torch.zeros(128, 128, 1, 1, device=cuda)) is None   # < 4096 pixels
    from univs_amd import layers
    conv = layers.Conv2d(Cin, Cout, kernel_size=1, bias=bias).to(cuda)
    with torch.no_grad():
        conv.weight.copy_(w)
        if bias:
            conv.bias.copy_(b)
        assert (conv(x) - ref32).abs().max().item() < max(8.0 * err32, 1e-5)


def test_presplit_weights_cache_and_wide_linear(cuda):
    """ops.presplit_weights: one split per weight tensor, redone after an in-place update; the wide-K Linear on the
    streamed three-product kernel == the six-product kernel to fp32 rounding, all epilogues, ragged rows, a short last pass."""
    from univs_amd.switches import override
    F = torch.nn.functional
    M, K, N = 5003, 1024, 208
    x = synth.normal("ps/x", (M, K)).to(cuda)
    w = synth.normal("ps/w", (N, K), std=K ** -0.5).to(cuda)
    b = synth.normal("ps/b", (N,)).to(cuda)
    r = synth.normal("ps/r", (M, N)).to(cuda)
    wp1, winv1 = ops.presplit_weights(w)
    wp2, _ = ops.presplit_weights(w)
    assert wp1 is wp2 and wp1.numel() == N * K and winv1.shape == (N,)
    ref64 = F.linear(x.double(), w.double(), b.double())
    ref32 = F.linear(x, w, b)
    e32 = (ref32.double() - ref64).abs().max().item()
    for act, res in ((None, None), ("relu", None), ("gelu", None), (None, r)):
        y = ops.linear_fused(x, w, b, act=act, residual=res)
        with override(presplit_kmin=0):
            assert ops.linear_fused(x, w, b, act=act, residual=res) is None      # K > 768 needs the pre-split weights
        want = ref64
        want = want.relu() if act == "relu" else F.gelu(want) if act == "gelu" else want
        want = want + res.double() if res is not None else want
        assert y is not None and (y.double() - want).abs().max().item() < max(4 * e32, 5e-6), act
    with torch.no_grad():
        w.mul_(2.0)                                                  # in-place: the version counter moves
    wp3, winv3 = ops.presplit_weights(w)
    assert wp3 is not wp1
    assert torch.equal(winv3, 2.0 * winv1) and torch.equal(wp3, wp1)  # a power of two only changes the row scales
    y2 = ops.linear_fused(x, w, None)
    assert (y2.double() - 2.0 * F.linear(x.double(), w.double() / 2)).abs().max().item() < max(8 * e32, 1e-5)
    with pytest.raises(RuntimeError):
        ops.presplit_weights(torch.zeros(4, 4, 3, 2, device=cuda), conv=True)


def test_presplit_weights_made_on_another_stream(cuda):
    """A weight split on a side stream (the prompt sampler's annotation work runs Linears there) and used right away on the
    current stream: the cache orders the consumer behind the split (an event per entry, waited for once per stream)."""
    F = torch.nn.functional
    M, K, N = 4096, 1024, 256
    x = synth.normal("pss/x", (M, K)).to(cuda)
    w = synth.normal("pss/w", (N, K), std=K ** -0.5).to(cuda)
    ref64 = F.linear(x.double(), w.double())
    e32 = (F.linear(x, w).double() - ref64).abs().max().item()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=cuda)
    with torch.cuda.stream(side):
        big = torch.empty(64 << 20, device=cuda).normal_()          # work in front of the split on the side stream
        wp, _ = ops.presplit_weights(w)
    y = ops.linear_fused(x, w, None)
    wp2, _ = ops.presplit_weights(w)
    torch.cuda.synchronize()
    assert wp2 is wp and y is not None and (y.double() - ref64).abs().max().item() < max(4 * e32, 5e-6)
    del big


@pytest.mark.parametrize("M,K,N,act", [(4100, 960, 72, None), (3000, 864, 136, "relu"), (2500, 2304, 264, "gelu"), (70000, 768, 192, None)],
                         ids=lambda v: str(v))
def test_streamed_linear_ring_of_three_and_ragged_shapes(cuda, M, K, N, act):
    """gemm_f16x3_stream beyond the model's own shapes: K a multiple of 96 but not of 128 (register ring / W groups of three
    k-steps), feature counts that leave a short last pass or a partial 16-feature block, ragged row tiles, idle waves."""
    F = torch.nn.functional
    x = synth.normal(f"sl/x/{M}x{K}", (M, K))
    x *= torch.exp2(torch.arange(M).remainder(9).float() - 4.0)[:, None]                  # rows over eight binades
    w = synth.normal(f"sl/w/{N}x{K}", (N, K), std=K ** -0.5)
    b = synth.normal(f"sl/b/{N}", (N,), std=0.5)
    xd, wd, bd = x.to(cuda), w.to(cuda), b.to(cuda)
    y = ops.linear_fused(xd, wd, bd, act=act)
    assert y is not None and tuple(y.shape) == (M, N)
    ref64 = F.linear(xd.double(), wd.double(), bd.double())
    ref32 = F.linear(xd, wd, bd)
    if act == "relu":
        ref64, ref32 = ref64.relu(), ref32.relu()
    if act == "gelu":
        ref64, ref32 = F.gelu(ref64), F.gelu(ref32)
    err = (y.double() - ref64).abs().max().item()
    err32 = (ref32.double() - ref64).abs().max().item()
    assert err < max(4.0 * err32, 5e-6), (err, err32)


@pytest.mark.parametrize("shape", [(5, 256, 184, 320), (3, 7, 16, 24), (1, 1, 8, 8), (2, 3, 272, 480)], ids=lambda v: str(v))
def test_bilinear_pyramid3_equals_three_resamplings(cuda, shape):
    """ops.bilinear_pyramid3 == three ops.bilinear_resample calls, bit for bit (and hence F.interpolate to rounding); sizes that
    are not multiples of 8 return None."""
    x = synth.normal(f"pyr/{shape}", shape).to(cuda)
    H, W = shape[-2:]
    pyr = ops.bilinear_pyramid3(x)
    assert pyr is not None and len(pyr) == 3
    for k, got in zip((2, 4, 8), pyr):
        want = ops.bilinear_resample(x, (H // k, W // k))
        assert torch.equal(got, want), k
        ref = torch.nn.functional.interpolate(x, size=(H // k, W // k), mode="bilinear", align_corners=False)
        assert (got - ref).abs().max().item() < 1e-6
    assert ops.bilinear_pyramid3(torch.zeros(2, 3, 12, 16, device=cuda)) is None


def test_layer_norm_second_output(cuda):
    """ops.layer_norm(..., post_add=p) == (LayerNorm(x + r), LayerNorm(x + r) + p): the encoder's norm2 and the next layer's
    `src + pos` from one pass."""
    x = synth.normal("ln2/x", (3, 1000, 256)).to(cuda)
    r = synth.normal("ln2/r", (3, 1000, 256)).to(cuda)
    p = synth.normal("ln2/p", (3, 1000, 256)).to(cuda)
    w = synth.normal("ln2/w", (256,)).to(cuda)
    b = synth.normal("ln2/b", (256,)).to(cuda)
    out, out2 = ops.layer_norm(x, w, b, 1e-5, residual=r, post_add=p)
    ref = ops.layer_norm(x, w, b, 1e-5, residual=r)
    assert torch.equal(out, ref) and torch.equal(out2, ref + p)
    out, out2 = ops.layer_norm(x, w, b, 1e-5, residual=r, post_add=p[:1])          # [1, S, C] repeated over the batch
    assert torch.equal(out, ref) and torch.equal(out2, ref + p[:1])
    with pytest.raises(RuntimeError):
        ops.layer_norm(x, w, b, 1e-5, post_add=p)
    with pytest.raises(RuntimeError):
        ops.layer_norm(x, w, b, 1e-5, residual=r, post_add=p[:, :1])                  # broadcast over rows: not covered


def test_conv3x3_channels_last_operand_is_bit_identical(cuda):
    """ops.conv3x3_nhwc (univs_conv3x3_nhwc_presplit_f32: the FPN convolution reading a channels-last operand, XMODE 2 of the streamed
    kernel) == ops.conv3x3 on the NCHW tensor bit for bit (same weights image, same k order), borders included; uncovered shapes -> None."""
    T, C, H, W = 2, 128, 46, 80
    x = synth.normal("c3n/x", (T, C, H, W)).to(cuda)
    w = synth.normal("c3n/w", (64, C, 3, 3), std=1 / 34).to(cuda)
    want = ops.conv3x3(x, w)
    got = ops.conv3x3_nhwc(x.permute(0, 2, 3, 1).contiguous(), w)
    assert want is not None and got is not None and torch.equal(got, want)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    assert (got.double() - ref).abs().max().item() < 5e-6 * ref.abs().max().item() + 1e-6
    assert ops.conv3x3_nhwc(x.permute(0, 2, 3, 1), w) is None                       # not contiguous as [T, H, W, C]
    assert ops.conv3x3_nhwc(x[:, :96].permute(0, 2, 3, 1).contiguous(), w[:, :96].contiguous()) is None   # Cin % 128


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,act,res", [(18400, 1536, 384, None, True), (4600, 3072, 768, None, True), (18400, 384, 384, None, True),
                                           (4600, 768, 2304, None, False), (4600, 768, 3072, "gelu", False), (18400, 768, 384, None, False),
                                           (4600, 1536, 768, None, False), (18399, 768, 200, "relu", False), (2049, 384, 132, None, True), (4100, 576, 260, None, False), (4100, 1152, 256, None, True),
                                           (73600, 384, 192, None, False)], ids=lambda v: str(v))
def test_linear_tile_kernel_is_bit_identical_to_the_pass_kernel(cuda, M, K, N, act, res):
    """The wide-K Linear tiled in two dimensions (gemm_f16x3_tile.hip: x split once per workgroup and shared through LDS) == the row-range x
    pass kernel (gemm_f16x3_stream.hip: every pass re-reads and re-splits x), bit for bit -- the same scales (running row maxima), the same
    parts and the same products in the same order per output; UnivsConfig.linear_ablate = 6 switches the tiled kernel off."""
    x = synth.normal(f"gt/x/{M}x{K}", (M, K)).to(cuda)
    x = x * torch.logspace(-2, 2, K, device=cuda).view(1, K)                      # the running row scale has to move along k
    w = (synth.normal(f"gt/w/{N}x{K}", (N, K), std=K ** -0.5) * torch.logspace(-3, 3, N).view(N, 1)).to(cuda)
    b = synth.normal(f"gt/b/{N}", (N,)).to(cuda)
    r = synth.normal(f"gt/r/{M}x{N}", (M, N)).to(cuda) if res else None
    with ops.configured(linear_ablate=6):
        want = ops.linear_fused(x, w, b, act=act, residual=r)
    got = ops.linear_fused(x, w, b, act=act, residual=r)
    assert want is not None and got is not None
    assert torch.equal(got, want)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    if act == "gelu":
        ref = torch.nn.functional.gelu(ref)
    if act == "relu":
        ref = ref.relu()
    if res:
        ref = ref + r.double()
    scale = (x.double().abs() @ w.double().abs().t()).max().item()
    assert (got.double() - ref).abs().max().item() < 2e-6 * scale


@pytest.mark.parametrize("Qp,L,T", [(10, 896, 5), (3, 1, 2), (1, 0, 1), (7, 130, 3)], ids=str)
def test_proca_attention_matches_the_reference_sequence(cuda, Qp, L, T):
    """univs_proca_attention_f32 (csrc/proca_attn.hip) == the reference's ProCA attention (…decoder_univs.py:456-496 ->
    nn.MultiheadAttention): memory = [state; dense tokens] per (prompt query, frame), q k^T / sqrt(d), softmax, p v -- here from
    already projected operands, against torch in float64."""
    E, h = 256, 8
    qkv0 = synth.normal(f"proca/qkv{Qp}", (Qp * T, 3 * E)).to(cuda)
    kd = synth.normal(f"proca/kd{Qp}", (Qp, L, T, E)).to(cuda)
    vd = synth.normal(f"proca/vd{Qp}", (Qp, L, T, E)).to(cuda)
    got = ops.proca_attention(qkv0, kd, vd, h)
    assert got is not None and tuple(got.shape) == (Qp * T, E)
    q, k0, v0 = qkv0.double().view(Qp, T, 3, h, 32).unbind(2)                              # [Qp, T, h, d]
    k = torch.cat([k0[:, None], kd.double().view(Qp, L, T, h, 32)], 1)                   # [Qp, 1 + L, T, h, d]
    v = torch.cat([v0[:, None], vd.double().view(Qp, L, T, h, 32)], 1)
    s = torch.einsum("qthd,qlthd->qthl", q, k) / 32 ** 0.5
    want = torch.einsum("qthl,qlthd->qthd", s.softmax(-1), v).reshape(Qp * T, E)
    assert (got.double() - want).abs().max().item() < 2e-5
    with pytest.raises(RuntimeError):
        ops.proca_attention(qkv0[:, :-1], kd, vd, h)
