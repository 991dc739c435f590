"""VisualPromptEncoder.get_mask_prompts (all key frames of a clip in one pass) against get_mask_prompt called once per
key frame (the restatement of prompt_encoder.py:168-263 that the goldens g7 / g11a / g15 / g20 pin): same tokens, same
attention masks, same sampled pixels, with the reference's host draws, with replayed draws and -- as far as a different
consumption order of the device generator allows -- with device draws."""
import pytest
import torch

from tests import cases, helpers


def _setup(n_ent=4, first_frame_idx=3):
    case = cases.HEAD_CASE
    head = helpers.build_head(case, "cpu", return_aux=False)
    enc = head.predictor.visual_prompt_sampler.visual_prompt_encoder
    tv = cases.targets_with_entities(case, first_frame_idx=first_frame_idx, n_ent=n_ent)[0]
    tv["masks"][1, 1] = 0                                     # an entity that is empty in one key frame
    tv["masks"][2, 0] = 0
    tv["masks"][2, 0, 40:48, 48:64] = 1.0                     # and one with fewer feature pixels than dense tokens
    Fk = enc.num_frames
    assert Fk == 2
    masks = tv["masks"][:, :Fk].transpose(0, 1).contiguous()  # [F, n, h, w]
    boxes = tv["boxes"][:, :Fk].transpose(0, 1).contiguous()
    s = enc.img_feats_scale
    h_img, w_img = masks.shape[-2] // s, masks.shape[-1] // s
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(Fk, 256, h_img, w_img, generator=g)
    pos = torch.randn(Fk, 256, h_img, w_img, generator=g)
    return enc, masks, boxes, feats, pos, h_img, w_img


def _per_frame(enc, masks, boxes, feats, pos, key_fids, kfos):
    outs = [enc.get_mask_prompt(feats[f], pos[f], masks=masks[f], boxes=boxes[f], key_fid=key_fids[f],
                                key_fid_original=kfos[f]) for f in range(masks.shape[0])]
    return [torch.stack([o[i] for o in outs]) for i in range(4)]


@pytest.mark.parametrize("mode", ["reference", "replay"])
def test_batched_key_frames_equal_one_call_per_frame(mode):
    enc, masks, boxes, feats, pos, h_img, w_img = _setup()
    key_fids, kfos = [0, 1], [7, 8]
    log_a, log_b = [], []
    with torch.no_grad():
        if mode == "replay":
            torch.manual_seed(11)
            enc.draw_log = rec = []
            _per_frame(enc, masks, boxes, feats, pos, key_fids, kfos)
            enc.set_replay(rec)
        enc.draw_log = log_a
        torch.manual_seed(3)
        ref = _per_frame(enc, masks, boxes, feats, pos, key_fids, kfos)
        if mode == "replay":
            assert enc.replay_pending() == 0
            enc.set_replay(rec)
        enc.draw_log = log_b
        torch.manual_seed(3)
        pre = enc.annotation_prefix(masks, boxes, h_img, w_img)
        counts = pre["counts"].tolist() if mode == "reference" else None
        got = enc.get_mask_prompts(feats, pos, masks, boxes, key_fids, kfos, pre, counts)
        enc.draw_log = None
        enc.set_replay(None)
    assert len(log_a) == len(log_b) == 2
    for (pa, fa), (pb, fb) in zip(log_a, log_b):
        assert torch.equal(pa, pb) and torch.equal(fa, fb)
    assert (log_a[1][1][1] == -1).all() and (log_a[0][1][1] >= 0).all()        # the empty entity of key frame 1
    assert torch.equal(got[0], ref[0])                                           # sampled points
    assert torch.equal(got[3], ref[3])                                           # attention masks
    assert (got[1] - ref[1]).abs().max().item() < 1e-6                           # position tokens
    assert (got[2] - ref[2]).abs().max().item() < 1e-5                           # feature tokens (bmm vs mm)
    assert got[2].shape == ref[2].shape and got[2].abs().sum() > 0


def test_batched_key_frames_with_device_draws():
    """device generator: the draws are consumed in another order than by separate calls, so only what does not depend on
    the stream is compared: sampled pixels lie on their masks, small / empty masks follow the deterministic rules."""
    enc, masks, boxes, feats, pos, h_img, w_img = _setup()
    old = enc.sampler_rng
    enc.sampler_rng = "device"
    try:
        with torch.no_grad():
            enc.draw_log = log = []
            pre = enc.annotation_prefix(masks, boxes, h_img, w_img)
            got = enc.get_mask_prompts(feats, pos, masks, boxes, [0, 1], [7, 8], pre, None)
            ref = _per_frame(enc, masks, boxes, feats, pos, [0, 1], [7, 8])
    finally:
        enc.sampler_rng, enc.draw_log = old, None
    assert torch.equal(got[3], ref[3])
    Fk, n = masks.shape[:2]
    fmb = pre["feat_masks_binary"].flatten(2)
    for f in range(Fk):
        pidx, fidx = log[f]
        for e in range(n):
            if fmb[f, e].sum() == 0:
                assert (fidx[e] == -1).all()
                continue
            assert fmb[f, e][fidx[e].long()].all(), (f, e)
            if pre["sel"][f, e].any():
                assert pre["sel"][f, e].flatten()[pidx[e].long()], (f, e)
    small = fmb[0, 2].sum().item()
    assert 0 < small < enc.num_dense_points, small
    assert torch.equal(log[0][1][2][:small].long(), torch.nonzero(fmb[0, 2]).flatten())    # cyclic fill, pixel order
