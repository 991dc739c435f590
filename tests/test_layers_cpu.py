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

    class FakeCuda:                     # a stand-in with the three attributes the note reads
        is_cuda = True
        shape = (300, 100)

        def numel(self):
            return 300 * 100
    w = torch.zeros(7, 100)
    layers._LIBRARY_FALLBACKS.clear()
    with caplog.at_level(logging.INFO, logger="univs_amd"):
        layers._note_library_linear(FakeCuda(), w)
        layers._note_library_linear(FakeCuda(), w)
        layers._note_library_linear(torch.zeros(3, 100), w)
    recs = [r for r in caplog.records if r.name == "univs_amd"]
    assert len(recs) == 1 and "100 -> 7" in recs[0].getMessage()
