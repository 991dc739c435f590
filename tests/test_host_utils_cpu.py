"""Host-side helpers: rank -> pixel search of the prompt sampler, mask / box utilities of the clip loop, runtime."""
import torch

from univs_amd import runtime, synth
from univs_amd.modeling.prompt_encoder import _kth_true, _kth_true_2d
from univs_amd.utils import comm


def test_kth_true_equals_nonzero_indexing():
    """`nonzero(mask[k])[rank]` for all rows at once (1-D scan and the two-level image version), incl. single-pixel
    rows, full rows and ranks at both ends."""
    m = synth.uniform("kth/m", (6, 23, 31)) > 0.4
    m[1] = False
    m[1, 22, 30] = True            # one pixel, the last one
    m[2] = True                    # every pixel
    m[3] = False
    m[3, 0, 0] = True              # the first one
    cnt = m.flatten(1).sum(1)
    g = torch.Generator().manual_seed(3)
    ranks = torch.stack([torch.cat([torch.tensor([0, int(c) - 1]), torch.randint(0, int(c), (5,), generator=g)]) for c in cnt])
    ref = torch.stack([torch.nonzero(m[k].flatten()).reshape(-1)[ranks[k]] for k in range(len(m))])
    assert torch.equal(_kth_true(m.flatten(1), ranks), ref)
    assert torch.equal(_kth_true_2d(m, ranks), ref)
    assert torch.equal(_kth_true_2d(m, ranks, m.sum(2, dtype=torch.int32)), ref)


def test_convert_mask_to_box_and_quality():
    m = torch.zeros(2, 3, 8, 10, dtype=torch.bool)
    m[0, 0, 2:5, 3:9] = True
    m[1, 2, 7, 0] = True
    b = comm.convert_mask_to_box(m)
    assert b[0, 0].tolist() == [3, 2, 8, 4] and b[1, 2].tolist() == [0, 7, 0, 7]
    assert b[0, 1].tolist() == [0, 0, 0, 0]                       # empty mask -> zeros
    assert tuple(comm.convert_mask_to_box(torch.zeros(0, 2, 4, 4, dtype=torch.bool)).shape) == (0, 2, 4)
    logit = torch.full((1, 2, 4, 4), -3.0)
    logit[0, :, :2] = 2.0           # 16 pixels above +1
    logit[0, :, 2, :2] = 0.0        # 4 pixels in the uncertain band
    assert abs(float(comm.calculate_mask_quality_scores(logit)) - 16.0 / 20.0) < 1e-6


def test_mask_and_box_iou():
    a = torch.zeros(1, 2, 4, 4)
    a[0, 0, :2] = 1
    a[0, 1, :, :2] = 1
    iou = comm.batched_mask_iou(a, a)
    assert torch.allclose(iou[0], torch.tensor([[1.0, 4.0 / 12.0], [4.0 / 12.0, 1.0]]))
    assert float(comm.batched_mask_iou(torch.zeros(1, 1, 2, 2), torch.zeros(1, 1, 2, 2))) == 0.0   # union clamped to 1
    b1 = torch.tensor([[[0.0, 0.0, 2.0, 2.0]]])
    b2 = torch.tensor([[[1.0, 1.0, 3.0, 3.0]], [[5.0, 5.0, 6.0, 6.0]]])
    iou, inter, union = comm.video_box_iou(b1, b2)
    assert torch.allclose(iou[0, :, 0], torch.tensor([1.0 / 7.0, 0.0]))


def test_runtime_gemm_table_is_inert_without_gpu():
    assert runtime.default_table() is not None and runtime.default_table().endswith(".csv")
    if not torch.cuda.is_available():
        assert runtime.enable_tuned_gemms() == "tunableop: no GPU"


def test_level_table_is_never_cached_and_is_validated():
    """a host copy of spatial_shapes keyed on (address, version) returned the previous
    resolution's table for freshly allocated tensors.  There is no cache, and inconsistent tables raise."""
    import pytest
    import torch
    from univs_amd import ops
    seen = []
    for rep in range(50):
        for shapes in ([[16, 6], [2, 3]], [[8, 6], [2, 3]], [[5, 7], [3, 2]]):
            starts = [0, shapes[0][0] * shapes[0][1]]
            S = starts[1] + shapes[1][0] * shapes[1][1]
            sh_t = torch.as_tensor(shapes, dtype=torch.long)       # fresh tensors every call, like the reference
            st_t = torch.as_tensor(starts, dtype=torch.long)
            sh, st, L = ops._host_shapes(sh_t, st_t, S)
            seen.append((list(sh), list(st)))
            assert list(sh) == [v for hw in shapes for v in hw] and list(st) == starts and L == 2
            del sh_t, st_t
    with pytest.raises(RuntimeError):   # a smaller table that merely fits inside S
        ops._host_shapes([[8, 6], [2, 3]], [0, 30], 102)
    with pytest.raises(RuntimeError):   # start index is not the running sum
        ops._host_shapes([[8, 6], [2, 3]], [0, 50], 54)
    with pytest.raises(RuntimeError):
        ops._host_shapes([[8, 6], [0, 3]], [0, 48], 48)


def test_inference_only_operators_refuse_to_drop_gradients():
    """the HIP operators return tensors detached from autograd; with gradient recording on
    they raise instead, and layers.linear keeps the ATen path so that autograd sees the Linear."""
    import pytest
    import torch
    from univs_amd import layers, ops
    x = torch.randn(4, 8, requires_grad=True)
    w = torch.ones(8)
    with pytest.raises(RuntimeError, match="inference-only"):
        ops.layer_norm(x, w, w)
    with torch.no_grad(), pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        ops.layer_norm(x, w, w)          # no_grad: passes the guard, then the CPU check of the operator
    assert ops.needs_grad(x) and not ops.needs_grad(x.detach())
    lin = torch.nn.Linear(256, 256)
    big = torch.randn(4096, 256, requires_grad=True)
    assert ops.linear_split(big, lin.weight, lin.bias) is None      # needs grad -> caller keeps F.linear
    y = layers.linear(big, lin.weight, lin.bias)
    y.sum().backward()
    assert big.grad is not None and lin.weight.grad is not None


def test_round2_operator_wrappers_on_cpu_tensors():
    """The fused Linear / convolution wrappers hand CPU tensors back to their callers (None = "keep your own Linear"),
    the transpose helper falls back to ATen, and the modules built on them give the ATen result on the CPU."""
    import torch

    from oracle import cpu_path
    from univs_amd import ops
    from univs_amd.modeling.backbone.swin import Mlp
    x = torch.randn(4, 2048, 96)
    w = torch.randn(288, 96)
    assert ops.linear_fused(x, w, None) is None
    assert ops.linear_fused(x, w, None, act="gelu") is None
    assert ops.conv3x3(torch.randn(1, 128, 64, 64), torch.randn(128, 128, 3, 3)) is None
    t = torch.randn(2, 8, 12)
    assert torch.equal(ops.transpose_last2(t), t.transpose(1, 2).contiguous())
    mlp = Mlp(96, 384).eval()
    with torch.no_grad():
        want = x + mlp.fc2(torch.nn.functional.gelu(mlp.fc1(x)))
        assert torch.equal(mlp(x, residual=x), want)
        # the oracle stand-in of the LayerNorm with a second output
        g, b, r, p = torch.randn(96), torch.randn(96), torch.randn(4, 2048, 96), torch.randn(1, 2048, 96)
        out, out2 = cpu_path.layer_norm(x, g, b, 1e-5, residual=r, post_add=p)
        ref = torch.nn.functional.layer_norm(x + r, (96,), g, b, 1e-5)
        assert torch.equal(out, ref) and torch.equal(out2, ref + p)


def test_fp32_region_leaves_autocast_and_upcasts_features_but_not_targets(monkeypatch):
    """Boundary B1 under the reference's `with autocast():` (train_net.py:334): module entries decorated with
    layers.fp32_region up-cast half-precision feature arguments (tensors, lists, dicts) and hand the caller-owned
    `targets` list through by identity (the prompt memory pool is mutated in place).  The autocast state is faked here;
    tests/test_modules_gpu.py runs the real thing."""
    import torch

    from univs_amd import layers

    class M:
        @layers.fp32_region
        def f(self, x, targets, extra=None):
            return x, targets, extra
    t = [{"prompt_feats": torch.ones(1, dtype=torch.half), "task": "detection"}]
    x, tt, e = M().f(torch.ones(2, dtype=torch.half), t, extra=torch.ones(1, dtype=torch.half))
    assert x.dtype == torch.half and tt is t and e.dtype == torch.half                     # autocast off: untouched
    monkeypatch.setattr(torch, "is_autocast_enabled", lambda *a: True)
    x, tt, e = M().f(torch.ones(2, dtype=torch.half), t, extra={"res2": [torch.ones(1, dtype=torch.bfloat16)]})
    assert x.dtype == torch.float32 and e["res2"][0].dtype == torch.float32
    assert tt is t and tt[0]["prompt_feats"].dtype == torch.half
    x, tt, _ = M().f(torch.ones(2, dtype=torch.int64), targets=t)
    assert x.dtype == torch.int64 and tt is t
