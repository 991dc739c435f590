"""ResNet backbone in Detectron2's module / state-dict layout (config 1: `MODEL.BACKBONE.NAME:
"build_resnet_backbone"`, configs/univs_inf/vids/Base.yaml:2-15: depth 50, STRIDE_IN_1X1 False, FrozenBN).

PARITY UNPINNED: the implementation the reference uses lives in detectron2 (`detectron2.modeling.backbone.
resnet.build_resnet_backbone`, un-pinned git HEAD per INSTALL.md:30-32) and is not part of /root/reference, and
no reference test pins its outputs.  This module follows the published structure
(torchvision / Detectron2 bottleneck ResNet: 7x7/2 stem + 3x3/2 max-pool, stages res2..res5 of
[3,4,6,3] bottlenecks, stride on the 3x3 conv, frozen BatchNorm as an affine transform) and the
Detectron2 parameter names (`stem.conv1.weight`, `stem.conv1.norm.{weight,bias,running_mean,running_var}`,
`res2.0.shortcut...`, `res2.0.conv{1,2,3}...`) so Detectron2 checkpoints load; tests check shapes, strides
and the key layout only.  Config-1 parity is asserted from `res2..res5` onward.
"""
import torch
import torch.nn.functional as F
from torch import nn

from ...layers import Conv2d, fp32_region
from ...registry import BACKBONE_REGISTRY, ShapeSpec


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm with fixed statistics: y = x * scale + shift (eps 1e-5), buffers named as in Detectron2."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def forward(self, x):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        bias = self.bias - self.running_mean * scale
        return x * scale.reshape(1, -1, 1, 1).to(x.dtype) + bias.reshape(1, -1, 1, 1).to(x.dtype)


class BottleneckBlock(nn.Module):
    def __init__(self, in_channels, out_channels, *, bottleneck_channels, stride=1, num_groups=1,
                 stride_in_1x1=False, dilation=1):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        if in_channels != out_channels:
            self.shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, bias=False,
                                   norm=FrozenBatchNorm2d(out_channels))
        else:
            self.shortcut = None
        stride_1x1, stride_3x3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.conv1 = Conv2d(in_channels, bottleneck_channels, kernel_size=1, stride=stride_1x1, bias=False,
                            norm=FrozenBatchNorm2d(bottleneck_channels))
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, kernel_size=3, stride=stride_3x3,
                            padding=1 * dilation, bias=False, groups=num_groups, dilation=dilation,
                            norm=FrozenBatchNorm2d(bottleneck_channels))
        self.conv3 = Conv2d(bottleneck_channels, out_channels, kernel_size=1, bias=False,
                            norm=FrozenBatchNorm2d(out_channels))

    def forward(self, x):
        out = F.relu_(self.conv1(x))
        out = F.relu_(self.conv2(out))
        out = self.conv3(out)
        shortcut = self.shortcut(x) if self.shortcut is not None else x
        return F.relu_(out + shortcut)


class BasicStem(nn.Module):
    def __init__(self, in_channels=3, out_channels=64):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, 4
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=7, stride=2, padding=3, bias=False,
                            norm=FrozenBatchNorm2d(out_channels))

    def forward(self, x):
        x = F.relu_(self.conv1(x))
        return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)


class ResNet(nn.Module):
    def __init__(self, stem, stages, out_features):
        super().__init__()
        self.stem = stem
        self._out_features = out_features
        self._out_feature_strides, self._out_feature_channels = {}, {}
        self.stage_names = []
        stride = stem.stride
        for i, blocks in enumerate(stages):
            name = f"res{i + 2}"
            self.add_module(name, nn.Sequential(*blocks))
            self.stage_names.append(name)
            stride *= blocks[0].stride
            self._out_feature_strides[name] = stride
            self._out_feature_channels[name] = blocks[-1].out_channels

    @fp32_region
    def forward(self, x):
        assert x.dim() == 4, f"ResNet takes an input of shape (N, C, H, W). Got {x.shape} instead!"
        outputs = {}
        x = self.stem(x)
        for name in self.stage_names:
            x = getattr(self, name)(x)
            if name in self._out_features:
                outputs[name] = x
        return outputs

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}

    @property
    def size_divisibility(self):
        return 32


@BACKBONE_REGISTRY.register()
def build_resnet_backbone(cfg, input_shape=None):
    r = cfg.MODEL.RESNETS
    depth = r.DEPTH
    blocks_per_stage = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}[depth]
    assert r.NORM in ("FrozenBN", ""), "only frozen BatchNorm (the shipped configs' default) is implemented"
    in_ch = input_shape.channels if input_shape is not None and input_shape.channels else 3
    stem = BasicStem(in_ch, r.STEM_OUT_CHANNELS)
    bottleneck = r.NUM_GROUPS * r.WIDTH_PER_GROUP
    in_channels, out_channels = r.STEM_OUT_CHANNELS, r.RES2_OUT_CHANNELS
    stages = []
    for idx, n in enumerate(blocks_per_stage):
        first_stride = 1 if idx == 0 else 2
        blocks = []
        for b in range(n):
            blocks.append(BottleneckBlock(in_channels, out_channels, bottleneck_channels=bottleneck,
                                          stride=first_stride if b == 0 else 1, num_groups=r.NUM_GROUPS,
                                          stride_in_1x1=r.STRIDE_IN_1X1))
            in_channels = out_channels
        stages.append(blocks)
        out_channels *= 2
        bottleneck *= 2
    return ResNet(stem, stages, out_features=r.OUT_FEATURES)
