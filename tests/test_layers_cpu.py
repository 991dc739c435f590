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
