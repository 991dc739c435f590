"""CPU: host-side routing of the Linear layers (univs_amd/layers.py).  The split-bf16 kernel is a GPU operator; on CPU tensors
`ops.linear_split` declines (None) and `layers.linear` / `linear_act` are exactly torch's functional forms."""
import torch
import torch.nn.functional as F

from univs_amd import layers, ops, synth


def test_linear_split_declines_cpu_and_uncovered_shapes():
    x = synth.normal("layers/x", (4096, 256))
    w = synth.normal("layers/w", (64, 256))
    assert ops.linear_split(x, w) is None                                   # CPU tensors: never a CPU implementation
    assert ops.linear_split(x.double(), w.double()) is None


def test_linear_and_linear_act_are_torch_on_cpu():
    x = synth.normal("layers/x2", (3, 700, 256))
    lin = torch.nn.Linear(256, 96)
    synth.load_synthetic(lin, "layers/lin.")
    with torch.no_grad():
        assert torch.equal(layers.linear(x, lin.weight, lin.bias), F.linear(x, lin.weight, lin.bias))
        assert torch.equal(layers.linear(x, lin.weight), F.linear(x, lin.weight))
        assert torch.equal(layers.linear_act(x, lin, F.relu), F.relu(lin(x)))
        assert torch.equal(layers.linear_act(x, lin, F.gelu), F.gelu(lin(x)))


def test_library_fallback_is_logged_once(caplog):
    """layers._note_library_linear: one INFO record per (K, N, rows bucket), none for CPU tensors (the GPU routing is in test_ops_gpu)."""
    import logging
    from univs_amd import layers

    class FakeCuda:                     # a stand-in with the attributes the note reads
        is_cuda = True
        shape = (300, 100)
        requires_grad = False
        dtype = torch.float32

        def numel(self):
            return 300 * 100
    w = torch.zeros(7, 100)
    layers._LIBRARY_FALLBACKS.clear()
    layers.reset_library_linear_counts()
    with caplog.at_level(logging.INFO, logger="univs_amd"):
        with torch.no_grad():
            layers._note_library_linear(FakeCuda(), w)
            layers._note_library_linear(FakeCuda(), w)
        layers._note_library_linear(torch.zeros(3, 100), w)
        half = FakeCuda()
        half.shape, half.dtype = (300, 64), torch.float16
        with torch.no_grad():
            layers._note_library_linear(half, torch.zeros(5, 64, dtype=torch.float16))
    recs = [r for r in caplog.records if r.name == "univs_amd"]
    assert len(recs) == 2 and "100 -> 7" in recs[0].getMessage() and "shape not covered" in recs[0].getMessage()
    assert "64 -> 5" in recs[1].getMessage() and "float16" in recs[1].getMessage()          # the actual reason is logged
    assert layers.LIBRARY_LINEAR_COUNTS == {("F.linear", 100, 7): 2, ("F.linear", 64, 5): 1}  # every call is counted


def test_packed_in_projection_rows_are_cached_whole_tensors_without_autograd():
    """MultiheadAttention._packed_rows: under no_grad the K / V rows of the packed in-projection are contiguous copies made once per
    weight version (whole tensors: the fp16 three-product Linear finds their split image in its cache), refreshed when the parameter
    changes in place or moves; with autograd the views of the parameter are returned so that gradients reach it."""
    import torch
    from oracle.cpu_path import cpu_ops
    from univs_amd.layers import MultiheadAttention
    torch.manual_seed(0)
    mha = MultiheadAttention(32, 4)
    with torch.no_grad():
        w1, b1 = mha._packed_rows(32, 32)
        w2, b2 = mha._packed_rows(32, 32)
        assert w1 is w2 and b1 is b2 and w1._base is None and w1.is_contiguous()
        assert torch.equal(w1, mha.in_proj_weight[32:64]) and torch.equal(b1, mha.in_proj_bias[32:64])
        mha.in_proj_weight.mul_(2.0)                              # an in-place update (load_state_dict, an optimizer step)
        w3, _ = mha._packed_rows(32, 32)
        assert w3 is not w1 and torch.equal(w3, mha.in_proj_weight[32:64])
        from univs_amd import ops
        mha.in_proj_weight.data.add_(1.0)                         # through `.data`: no version bump -> ops.invalidate_presplit() is the call
        assert mha._packed_rows(32, 32)[0] is w3
        ops.invalidate_presplit()
        w4, _ = mha._packed_rows(32, 32)
        assert w4 is not w3 and torch.equal(w4, mha.in_proj_weight[32:64])
        q = torch.randn(5, 2, 32)
        k = torch.randn(7, 2, 32)
        with cpu_ops():
            out_cached = mha(q, k, k + 1.0)[0]
    wv, bv = mha._packed_rows(32, 32)                             # autograd on: views of the parameters
    assert wv._base is not None and wv.requires_grad
    with cpu_ops():
        out_views = mha(q, k, k + 1.0)[0]
    assert torch.allclose(out_cached, out_views.detach(), atol=1e-6)
    out_views.sum().backward()
    assert mha.in_proj_weight.grad is not None and mha.in_proj_weight.grad[32:].abs().sum() > 0
