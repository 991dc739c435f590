"""Builders resolving the reference's registry names from a cfg (detectron2 `build_backbone` /
`build_sem_seg_head` equivalents) and the hot-path model shell."""
import torch
from torch import nn

from ..registry import BACKBONE_REGISTRY, SEM_SEG_HEADS_REGISTRY, ShapeSpec
# importing the modules registers the classes
from .backbone import resnet as _resnet  # noqa: F401
from .backbone import swin as _swin  # noqa: F401
from .meta_arch import mask_former_head as _head  # noqa: F401
from .pixel_decoder import msdeformattn as _pd  # noqa: F401
from .transformer_decoder import univs_decoder as _dec  # noqa: F401


def build_model(cfg):
    """detectron2's `build_model`: resolves cfg.MODEL.META_ARCHITECTURE ('UniVS_Prompt')."""
    from ..registry import META_ARCH_REGISTRY
    from .meta_arch import univs_prompt as _meta  # noqa: F401  (registers the class)
    return META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)


def build_backbone(cfg, input_shape=None):
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    return BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)


def build_sem_seg_head(cfg, input_shape):
    return SEM_SEG_HEADS_REGISTRY.get(cfg.MODEL.SEM_SEG_HEAD.NAME)(cfg, input_shape)


class UniVSHotPath(nn.Module):
    """backbone + sem_seg_head with the attribute names the reference's META_ARCH shells use
    (`model.backbone`, `model.sem_seg_head`; univs/univs_prompt.py:218-353), plus the caller-side
    normalise + pad step (univs/inference/inference_video_entity.py:251-260)."""

    def __init__(self, cfg):
        super().__init__()
        self.backbone = build_backbone(cfg)
        self.sem_seg_head = build_sem_seg_head(cfg, self.backbone.output_shape())
        self.register_buffer("pixel_mean", torch.tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1), False)
        self.size_divisibility = cfg.MODEL.MASK_FORMER.SIZE_DIVISIBILITY
        self.num_queries = cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES

    def preprocess(self, frames):
        """frames [T,3,H,W] in 0..255 -> normalised, zero-padded to a multiple of size_divisibility."""
        x = (frames - self.pixel_mean) / self.pixel_std
        d = self.size_divisibility
        H, W = x.shape[-2:]
        Hp, Wp = (H + d - 1) // d * d, (W + d - 1) // d * d
        if (Hp, Wp) != (H, W):
            x = torch.nn.functional.pad(x, (0, Wp - W, 0, Hp - H))
        return x

    @torch.no_grad()
    def forward(self, frames, targets):
        x = self.preprocess(frames)
        return self.sem_seg_head(self.backbone(x), targets=targets)
