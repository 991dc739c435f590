"""GPU: ops.mask_stats (csrc/mask_stats.hip, univs_mask_stats_f32) against the ATen formulation it replaces in the clip loop --
`calculate_mask_quality_scores`' two counts and `convert_mask_to_box` (univs_amd/utils/comm.py, themselves pinned against the reference by
the golden loop tests).  Integers: exact."""
import pytest
import torch

from univs_amd import ops
from univs_amd.utils.comm import convert_mask_to_box

pytestmark = pytest.mark.gpu


def reference(x, t_hi, t_lo, t_box, valid):
    cur = x if valid is None else x[..., : valid[0], : valid[1]]
    hi = (cur > t_hi).flatten(-2).sum(-1)
    lo = (cur > t_lo).flatten(-2).sum(-1)
    box = convert_mask_to_box(cur > t_box)
    ne = (cur > t_box).flatten(-2).any(-1)
    return torch.cat([hi[..., None], lo[..., None], box, ne[..., None].long(), torch.zeros_like(hi)[..., None]], -1)


@pytest.mark.parametrize("shape,valid", [((7, 3, 20, 28), None), ((7, 3, 20, 28), (17, 25)), ((5, 2, 19, 27), None), ((5, 2, 19, 27), (19, 26)),
                                         ((100, 5, 184, 320), None), ((3, 5, 736, 1280), (720, 1280)), ((1, 1, 8, 4), None),
                                         ((300, 1, 33, 64), (33, 61))])
def test_mask_stats_matches_aten(cuda, shape, valid):
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(shape, generator=g) * 2.0).to(cuda)
    flat = x.view(-1, shape[-2], shape[-1])
    flat[0] = -3.0                                     # an empty plane
    flat[-1, :, :] = -3.0
    flat[-1, shape[-2] // 2, shape[-1] - 1] = 5.0      # a single pixel in the last column
    if flat.shape[0] > 2:
        flat[1, 0, 0] = float("nan")
        flat[1, 1, 1] = float("inf")
        flat[2] = 4.0                                  # a full plane
    for th in ((1.0, -1.0, 0.0), (0.5, 0.5, 0.5)):
        got = ops.mask_stats(x, *th, valid=valid)
        assert got is not None and got.dtype == torch.int32 and tuple(got.shape) == tuple(shape[:-2]) + (8,)
        ref = reference(x, *th, valid)
        assert torch.equal(got.long(), ref), (shape, valid, th, (got.long() != ref).nonzero()[:5].tolist())
    # run to run: integer atomics, identical
    assert torch.equal(ops.mask_stats(x, valid=valid), ops.mask_stats(x, valid=valid))


def test_mask_stats_refuses_what_it_does_not_cover(cuda):
    with pytest.raises(RuntimeError):
        ops.mask_stats(torch.zeros(2, 4, 4), 1.0, -1.0, 0.0)                   # CPU tensor: no CPU path
    with pytest.raises(RuntimeError):
        ops.mask_stats(torch.zeros(2, 4, 4, device=cuda)[:, :, ::2])           # not contiguous
    assert ops.mask_stats(torch.zeros(70000, 2, 2, device=cuda)) is None        # more planes than the grid takes: the caller keeps ATen
    assert tuple(ops.mask_stats(torch.zeros(0, 3, 4, 4, device=cuda)).shape) == (0, 3, 8)


def test_mask_stats_reads_a_history_view_in_place(cuda):
    """`history[:, -T:]` ([N, T, H, W] with plane strides (hist * H * W, H * W)): the statistics of the view without a copy of it."""
    g = torch.Generator().manual_seed(11)
    hist = (torch.randn(6, 9, 20, 28, generator=g) * 2.0).to(cuda)
    hist[2, -2] = -3.0
    for T in (1, 4, 9):
        view = hist[:, -T:]
        assert not view.is_contiguous() or T == 9
        got = ops.mask_stats(view, 1.0, -1.0, 0.0, valid=(19, 27))
        assert torch.equal(got.long(), reference(view.contiguous(), 1.0, -1.0, 0.0, (19, 27))), T
    odd = (torch.randn(3, 5, 7, 9, generator=g)).to(cuda)[:, 1:4]                 # odd width: the scalar path on a view
    assert torch.equal(ops.mask_stats(odd).long(), reference(odd.contiguous(), 1.0, -1.0, 0.0, None))
