"""Differentiable CPU restatement of multi-scale deformable attention, for the BACKWARD parity tests.

TEST INFRASTRUCTURE (never imported by univs_amd/).  Follows the reference's own pure-PyTorch core
`ms_deform_attn_core_pytorch` (mask2former/modeling/pixel_decoder/ops/functions/ms_deform_attn_func.py:52-72),
which the reference uses as the oracle of its CUDA op, gradients included (ops/test.py:66-87
`check_gradient_numerical`): per level, `F.grid_sample(value_l, 2*loc - 1, bilinear, zeros,
align_corners=False)`, weighted by the attention weights and summed over levels and points.
Pinned against the real reference by tests/golden/g13_msda_backward.npz (tests/test_oracle_golden.py).
"""
import torch
import torch.nn.functional as F


def forward(value, spatial_shapes, sampling_locations, attention_weights):
    """value [N,S,M,D], sampling_locations [N,Lq,M,L,P,2] in [0,1], attention_weights [N,Lq,M,L,P]
    -> [N, Lq, M*D]"""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    shapes = [(int(h), int(w)) for h, w in (spatial_shapes.tolist() if isinstance(spatial_shapes, torch.Tensor) else spatial_shapes)]
    grids = 2 * sampling_locations - 1
    out = value.new_zeros((N * M, D, Lq))
    start = 0
    for l, (H, W) in enumerate(shapes):
        v = value[:, start:start + H * W].permute(0, 2, 3, 1).reshape(N * M, D, H, W)
        start += H * W
        g = grids[:, :, :, l].permute(0, 2, 1, 3, 4).reshape(N * M, Lq, P, 2)
        samp = F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False)   # [N*M, D, Lq, P]
        w = attention_weights[:, :, :, l].permute(0, 2, 1, 3).reshape(N * M, 1, Lq, P)
        out = out + (samp * w).sum(-1)
    return out.view(N, M * D, Lq).transpose(1, 2).contiguous()


def backward(value, spatial_shapes, sampling_locations, attention_weights, grad_output):
    """[grad_value, grad_sampling_loc, grad_attn_weight] by autograd through `forward`."""
    v = value.detach().clone().requires_grad_(True)
    loc = sampling_locations.detach().clone().requires_grad_(True)
    aw = attention_weights.detach().clone().requires_grad_(True)
    out = forward(v, spatial_shapes, loc, aw)
    out.backward(grad_output)
    return [v.grad, loc.grad, aw.grad]
