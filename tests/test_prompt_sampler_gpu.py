"""GPU: the visual-prompt sampler kernels (csrc/prompt_sampler.hip: univs_prompt_prefix_f32, univs_prompt_draw, univs_prompt_tokens_f32) against
the ATen formulation of the same steps (univs_amd/modeling/prompt_encoder.py with SWITCHES.fused_sampler off -- the CPU path, itself checked
against the reference's sampler in tests/test_prompt_encoder_cpu.py / test_sampler_device_cpu.py): bit for bit, in all three draw modes."""
import pytest
import torch

from univs_amd import synth
from univs_amd.modeling.prompt_encoder import VisualPromptEncoder
from univs_amd.switches import override

pytestmark = pytest.mark.gpu

R, S = 16, 8


def scene(Fk, n, hi, wi, dev, seed, S=S):
    """F key frames x n entities: blobs with soft edges (values in [0, 1]), one empty entity, one tiny one, one whose box misses its mask
    (no central pixel: the fallback to the most confident pixels), one below the validity threshold"""
    g = torch.Generator().manual_seed(seed)
    h, w = hi * S, wi * S
    masks = torch.zeros(Fk, n, h, w)
    boxes = torch.zeros(Fk, n, 4)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    for f in range(Fk):
        for e in range(n):
            kind = (e + f) % 6
            if kind == 0:
                continue                                                       # empty: zero mask, zero box
            cy, cx = float(torch.rand(1, generator=g)) * h, float(torch.rand(1, generator=g)) * w
            ry, rx = (1.4 * S, 1.9 * S) if kind == 1 else (h * (0.05 + 0.3 * float(torch.rand(1, generator=g))), w * (0.05 + 0.3 * float(torch.rand(1, generator=g))))
            d = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2
            m = (1.2 - d).clamp(0, 1)
            if kind == 2:
                m = m * 0.4                                                   # never above 0.5: invalid, but visible
            if kind == 3:
                m = m * 0.7                                                   # max below 0.75: the threshold is the maximum itself
            masks[f, e] = m
            on = m > 0.3
            if on.any():
                ys, xs = on.any(1).nonzero().flatten(), on.any(0).nonzero().flatten()
                boxes[f, e] = torch.tensor([xs[0] / w, ys[0] / h, (xs[-1] + 1) / w, (ys[-1] + 1) / h])
            if kind == 4:
                boxes[f, e] = torch.tensor([0.0, 0.0, 0.05, 0.05])             # a box that misses the mask
    C = 256
    feats = synth.normal(f"psamp/feats/{seed}", (Fk, C, hi, wi))
    pos = synth.normal(f"psamp/pos/{seed}", (Fk, hi, wi, C)).permute(0, 3, 1, 2)      # channels-last, as the position embeddings are
    return masks.to(dev), boxes.to(dev), feats.to(dev), pos.to(dev)


def encoder(mode, T=3, scale=S):
    enc = VisualPromptEncoder(hidden_dim=256, num_frames=T, num_dense_points=R, position_embedding_sin3d_type="ArbitraryT")
    enc.sampler_rng = mode
    enc.img_feats_scale = scale
    return enc


def run(enc, masks, boxes, feats, pos, fused, seed=11, replay=None):
    Fk, n = masks.shape[:2]
    hi, wi = feats.shape[-2:]
    with override(fused_sampler=fused), torch.no_grad():
        torch.manual_seed(seed)
        enc.begin_video(masks.device)
        enc.set_replay(replay)
        enc.draw_log = []
        pre = enc.annotation_prefix(masks, boxes, hi, wi)
        counts = None
        if enc._rng(masks.device) == "reference" and replay is None:
            counts = pre["counts"].tolist()
        out = enc.get_mask_prompts(feats, pos, masks, boxes, list(range(Fk)), [7 + k for k in range(Fk)], pre, counts)
        log, enc.draw_log = enc.draw_log, None
    return pre, out, log


@pytest.mark.parametrize("Fk,n,hi,wi,scale", [(2, 6, 16, 24, 8), (1, 7, 23, 40, 8), (2, 10, 92, 160, 8), (1, 6, 75, 101, 1), (2, 6, 30, 45, 3),
                                              (1, 8, 136, 240, 8)])     # (the last: BASELINE config 5's 1088 x 1920, 32 640 keys per entity in LDS)
@pytest.mark.parametrize("mode", ["device", "reference"])
def test_fused_sampler_equals_the_aten_formulation(cuda, Fk, n, hi, wi, scale, mode):
    """(scale 1 and 3: mask rows that are not multiples of 16 bytes -- the kernels' scalar path)"""
    masks, boxes, feats, pos = scene(Fk, n, hi, wi, cuda, seed=Fk * 100 + n, S=scale)
    enc = encoder(mode, scale=scale)
    pre_a, out_a, log_a = run(enc, masks, boxes, feats, pos, fused=False)
    pre_f, out_f, log_f = run(enc, masks, boxes, feats, pos, fused=True)
    assert sorted(pre_a) == sorted(pre_f)
    for k in pre_a:
        assert pre_a[k].shape == pre_f[k].shape and pre_a[k].dtype == pre_f[k].dtype, k
        assert torch.equal(pre_a[k], pre_f[k]), k
    # the scene holds every branch
    cnt = pre_f["counts"]
    assert (cnt[:, n:] == 0).any() and ((cnt[:, n:] > 0) & (cnt[:, n:] < R)).any() and (cnt[:, n:] >= R).any()
    assert (~pre_f["valid"] & pre_f["visible"]).any()
    for a, b, name in zip(out_a, out_f, ("point_coords", "pd", "fd", "attn")):
        assert a.shape == b.shape and a.dtype == b.dtype, name
        assert torch.equal(a, b), name
    assert len(log_a) == len(log_f) == Fk
    for (pa, da), (pf_, df) in zip(log_a, log_f):
        assert torch.equal(pa, pf_) and torch.equal(da, df)
    # replaying the recorded pixels through the fused token kernel gives the same tokens again
    enc_r = encoder(mode, scale=scale)
    _, out_r, _ = run(enc_r, masks, boxes, feats, pos, fused=True, replay=log_f)
    for a, b in zip(out_f, out_r):
        assert torch.equal(a, b)


def test_fused_sampler_draws_inside_the_masks_and_without_repeats(cuda):
    """properties that do not lean on the ATen formulation: the point is a candidate pixel, the R dense pixels of a large mask are
    distinct pixels of its binary feature mask, a small mask's pixels repeat cyclically in raster order"""
    Fk, n, hi, wi = 2, 10, 92, 160
    masks, boxes, feats, pos = scene(Fk, n, hi, wi, cuda, seed=5)
    enc = encoder("device")
    pre, out, log = run(enc, masks, boxes, feats, pos, fused=True)
    h, w = masks.shape[-2:]
    sel, fmb, cnt = pre["sel"].cpu(), pre["feat_masks_binary"].cpu().flatten(2), pre["counts"].cpu()
    for f, (pidx, didx) in enumerate(log):
        for e in range(n):
            if cnt[f, e] > 0:
                assert sel[f, e].flatten()[int(pidx[e])]
            d, c = didx[e].long(), int(cnt[f, n + e])
            if c == 0:
                assert (d == -1).all()
            elif c >= R:
                assert fmb[f, e][d].all() and d.unique().numel() == R
            else:
                on = fmb[f, e].nonzero().flatten()
                assert torch.equal(d, on[torch.arange(R) % c])


def test_prompt_kernels_reject_bad_arguments(cuda):
    from univs_amd import ops
    masks, boxes, feats, pos = scene(1, 3, 16, 24, cuda, seed=1)
    with pytest.raises(RuntimeError):
        ops.prompt_prefix(masks, boxes, 7)                     # the scale does not divide the mask
    with pytest.raises(RuntimeError):
        ops.prompt_prefix(masks.double(), boxes, 8)
    with pytest.raises(RuntimeError):
        ops.prompt_prefix(masks.cpu(), boxes.cpu(), 8)
    pre = ops.prompt_prefix(masks, boxes, 8)
    with pytest.raises(RuntimeError):
        ops.prompt_draw(pre, R, tab=torch.zeros(3, R, dtype=torch.int64, device=cuda))
    assert ops.prompt_prefix(masks[:0], boxes[:0], 8)["counts"].shape == (0, 6)


@pytest.mark.parametrize("kind", ["ArbitraryT", "FixedT"])
def test_point_position_tokens_are_the_aten_bits(cuda, kind):
    """ops.prompt_point_pe == the diagonal of position_encoding._points (every frame's z against every point), bit for bit"""
    from univs_amd import ops
    from univs_amd.modeling.position_encoding import _axis, _dim_t
    Fk, n, T = 3, 7, 5
    enc = VisualPromptEncoder(hidden_dim=256, num_frames=T, num_dense_points=R, position_embedding_sin3d_type=kind)
    pl = enc.pe_layer
    xy = torch.rand(Fk * n, 2, generator=torch.Generator().manual_seed(3)).to(cuda)
    xy[0] = 0.0
    xy[1] = 1.0
    if kind == "FixedT":
        z = _axis(T, pl.scale, cuda)[torch.tensor([0, 2, 4], device=cuda)]
    else:
        z = torch.tensor([0, 17, 127], device=cuda) / pl.num_max_frames * pl.scale
    ar = torch.arange(Fk, device=cuda)
    want = pl._points(z, xy).view(Fk, Fk, n, -1)[ar, ar].reshape(Fk * n, -1)
    got = ops.prompt_point_pe(xy, z, _dim_t(pl.num_pos_feats, pl.temperature, cuda), _dim_t(2 * pl.num_pos_feats, pl.temperature, cuda), pl.scale, n)
    assert got.shape == want.shape == (Fk * n, 256) and torch.equal(got, want)


def test_token_mean_counts_non_blank_tokens(cuda):
    """ops.token_mean == x.sum(1) / clamp(number of tokens that are not all zero, 1) (+ add): blank tokens (all channels zero) do not
    count, an entity without any token gives the bare `add`, a token with ONE non-zero channel counts"""
    from univs_amd import ops
    n, L, T, C = 5, 33, 3, 256
    x = synth.normal("tokmean/x", (n, L, T, C))
    x[0, 5:] = 0.0                                   # 5 tokens
    x[1] = 0.0                                       # none
    x[2, 7] = 0.0
    x[2, 7, 1, 200] = 3.0                            # one channel of one frame
    x[3, :, 2] = 0.0                                 # frame 2 blank for entity 3
    add = synth.normal("tokmean/add", (C,))
    xd, ad = x.to(cuda), add.to(cuda)
    for a in (ad, None):
        got = ops.token_mean(xd, a)
        nb = torch.logical_not((xd == 0).all(dim=-1)).unsqueeze(-1).sum(1).clamp(min=1)
        want = xd.sum(1) / nb
        if a is not None:
            want = want + a.view(1, 1, -1)
        assert got.shape == (n, T, C)
        assert (got - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item())
    assert torch.equal(ops.token_mean(xd, ad)[1], ad.view(1, C).expand(T, C))
    assert torch.equal(ops.token_mean(xd, None)[3, 2], torch.zeros(C, device=cuda))
    x64 = x.double()
    exact = x64.sum(1) / torch.logical_not((x64 == 0).all(-1)).unsqueeze(-1).sum(1).clamp(min=1)
    assert (ops.token_mean(xd).double().cpu() - exact).abs().max().item() < 5e-6
    assert ops.token_mean(x) is None                 # CPU tensors: the caller keeps the ATen formulation


def test_feature_maps_smaller_than_the_token_count_fall_back(cuda):
    """fewer feature pixels than dense tokens (3 x 5 < R = 16): `prompt_draw` declines and the ATen formulation turns the SAME draws into
    pixels; prefix and token kernels still run -- the clip's tensors equal the pure ATen run"""
    from univs_amd import ops
    Fk, n, hi, wi = 1, 4, 3, 5
    masks, boxes, feats, pos = scene(Fk, n, hi, wi, cuda, seed=77)
    pre = ops.prompt_prefix(masks, boxes, S)
    assert ops.prompt_draw(pre, R, u=torch.rand(Fk * n, 1, device=cuda), keys=torch.rand(Fk * n, hi * wi, device=cuda)) is None
    enc = encoder("device")
    _, out_a, log_a = run(enc, masks, boxes, feats, pos, fused=False)
    _, out_f, log_f = run(enc, masks, boxes, feats, pos, fused=True)
    for a, b in zip(out_a, out_f):
        assert torch.equal(a, b)
    assert all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(log_a, log_f))
