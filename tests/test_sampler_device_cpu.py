"""CPU: the opt-in device-side prompt sampler (UNIVS_SAMPLER=device, univs_amd/modeling/prompt_encoder.py).  Same
distributions as the reference's host `randperm` draws, a different random stream, and NO host round trip; everything that is
not random must equal the reference mode."""
import pytest
import torch

from univs_amd import synth
from univs_amd.modeling.prompt_encoder import VisualPromptEncoder

R, HF, WF, S = 16, 16, 24, 8          # dense tokens per frame, feature-map size, mask / feature stride


def encoders():
    ref = VisualPromptEncoder(hidden_dim=256, num_frames=2, num_dense_points=R, position_embedding_sin3d_type="ArbitraryT")
    dev = VisualPromptEncoder(hidden_dim=256, num_frames=2, num_dense_points=R, position_embedding_sin3d_type="ArbitraryT")
    assert ref.sampler_rng == "reference"
    dev.sampler_rng = "device"
    return ref, dev


def scene():
    masks = torch.zeros(4, HF * S, WF * S)
    masks[0, 16:100, 24:150] = 1.0            # large: more than R feature pixels
    masks[1, 40:56, 64:88] = 1.0              # small: 2 x 3 feature pixels (< R): cyclic fill
    masks[3, 8:120, 160:184] = 1.0            # tall and thin
    # entity 2 stays empty
    feats = synth.normal("sampler/feats", (256, HF, WF))
    feats[0] = torch.arange(HF * WF, dtype=torch.float32).view(HF, WF)          # channel 0 = pixel index
    pos = synth.normal("sampler/pos", (256, HF, WF))
    return masks, feats, pos


def test_device_mode_draws_inside_the_masks_without_host_round_trips(monkeypatch):
    ref, dev = encoders()
    masks, feats, pos = scene()
    torch.manual_seed(3)
    p_ref, pd_ref, fd_ref, am_ref = ref.get_mask_prompt(feats, pos, masks, key_fid=0, key_fid_original=5)

    def no_sync(self):
        raise AssertionError("host round trip (.tolist) in device sampler mode")
    monkeypatch.setattr(torch.Tensor, "tolist", no_sync)
    p_dev, pd_dev, fd_dev, am_dev = dev.get_mask_prompt(feats, pos, masks, key_fid=0, key_fid_original=5)
    monkeypatch.undo()

    assert torch.equal(am_dev, am_ref)                                        # attention masks do not depend on the draws
    assert fd_dev.shape == fd_ref.shape == (4, R, 2, 256) and pd_dev.shape == pd_ref.shape
    # empty entity: zero tokens in both modes; small entity: the cyclic rule, identical in both modes
    assert fd_dev[2].abs().max() == 0 and pd_dev[2].abs().max() == 0
    assert torch.equal(fd_dev[1], fd_ref[1]) and torch.equal(pd_dev[1], pd_ref[1])
    fm = torch.nn.functional.interpolate(masks[:, None], (HF, WF), mode="nearest")[:, 0] >= 0.5
    for e in (0, 3):                                                          # R distinct pixels of the feature mask
        idx = fd_dev[e, :, 0, 0].long()
        assert fm[e].flatten()[idx].all() and idx.unique().numel() == R
        assert torch.equal(fd_dev[e, :, 0], fd_dev[e, :, 1])                  # replicated over the clip's frames
    # the query point of every non-empty entity is one of its candidate pixels
    sel, _ = dev._select_candidates(masks, None)
    H, W = masks.shape[-2:]
    for e in (0, 1, 3):
        x, y = int(p_dev[e, 0] * W), int(p_dev[e, 1] * H)
        assert sel[e, y, x]


def test_device_mode_is_seeded_and_uniform():
    _, dev = encoders()
    masks = torch.zeros(1, HF * S, WF * S)
    masks[0, 32:64, 32:72] = 1.0                                              # 32 x 40 pixels, central half = candidates
    sel, rowcnt = dev._select_candidates(masks, None)
    n_cand = int(sel.sum())
    torch.manual_seed(11)
    a = dev.select_points_from_box_mask(HF, WF, masks=masks, num_points=4000)
    dev._dev_gen.clear()
    torch.manual_seed(11)
    b = dev.select_points_from_box_mask(HF, WF, masks=masks, num_points=4000)
    assert torch.equal(a, b)                                                  # seeded by torch.manual_seed
    H, W = masks.shape[-2:]
    pix = (a[0, :, 1] * H).long() * W + (a[0, :, 0] * W).long()
    assert sel[0].flatten()[pix].all()
    counts = torch.bincount(pix, minlength=H * W)[sel[0].flatten()]
    expect = 4000 / n_cand
    assert counts.numel() == n_cand and (counts.float() - expect).abs().max() < 6 * expect ** 0.5 + 1


def test_auto_mode_resolves_by_device():
    """The default ("auto"): the reference's host draws for CPU tensors, the device generator for GPU tensors."""
    from univs_amd.switches import Switches
    assert Switches().sampler == "auto"
    enc = VisualPromptEncoder(hidden_dim=256, num_frames=2, num_dense_points=R)
    enc.sampler_rng = "auto"
    assert enc._rng(torch.zeros(1)) == "reference" and enc._rng(torch.device("cpu")) == "reference"
    assert enc._rng(torch.device("cuda", 0)) == "device"
    enc.sampler_rng = "device"
    assert enc._rng(torch.zeros(1)) == "device"
    masks, feats, pos = scene()
    enc.sampler_rng = "auto"
    ref, _ = encoders()
    torch.manual_seed(3)
    a = enc.get_mask_prompt(feats, pos, masks, key_fid=0, key_fid_original=5)
    torch.manual_seed(3)
    b = ref.get_mask_prompt(feats, pos, masks, key_fid=0, key_fid_original=5)
    assert all(torch.equal(x, y) for x, y in zip(a, b))          # on the CPU "auto" IS the reference mode


def test_unknown_mode_is_rejected():
    from univs_amd.switches import override
    with override(sampler="fast"), pytest.raises(ValueError):
        VisualPromptEncoder(hidden_dim=256, num_frames=2, num_dense_points=R)


def test_begin_video_makes_a_seeded_video_reproducible():
    """VisualPromptEncoder.begin_video (called by the clip loops at the start of every video): in device mode the generators are
    reseeded from a value drawn from the default generator -- `torch.manual_seed(s)` in front of a video fixes its draws whatever
    ran before (the generator used to be seeded once per process); in reference mode nothing is drawn (the
    reference's random stream, draw for draw)."""
    ref, dev = encoders()
    masks = torch.zeros(1, HF * S, WF * S)
    masks[0, 32:64, 32:72] = 1.0
    torch.manual_seed(5)
    dev.begin_video(torch.device("cpu"))
    a = dev.select_points_from_box_mask(HF, WF, masks=masks, num_points=64)
    for _ in range(3):                                                       # other videos in between: the generator moves on
        dev.select_points_from_box_mask(HF, WF, masks=masks, num_points=17)
    torch.manual_seed(5)
    dev.begin_video(torch.device("cpu"))
    b = dev.select_points_from_box_mask(HF, WF, masks=masks, num_points=64)
    assert torch.equal(a, b)
    torch.manual_seed(6)
    dev.begin_video(torch.device("cpu"))
    c = dev.select_points_from_box_mask(HF, WF, masks=masks, num_points=64)
    assert not torch.equal(a, c)
    torch.manual_seed(9)
    before = torch.get_rng_state()
    assert ref.begin_video(torch.device("cpu")) is None and torch.equal(torch.get_rng_state(), before)
