"""`MaskFormerHead`: pixel decoder -> transformer predictor glue
(mask2former/modeling/meta_arch/mask_former_head.py:20-165; `layers` :148-154)."""
from typing import Dict

from torch import nn

from ...registry import SEM_SEG_HEADS_REGISTRY, TRANSFORMER_DECODER_REGISTRY, ShapeSpec, configurable


def build_pixel_decoder(cfg, input_shape):
    """mask2former/modeling/pixel_decoder/fpn.py:21-33"""
    name = cfg.MODEL.SEM_SEG_HEAD.PIXEL_DECODER_NAME
    model = SEM_SEG_HEADS_REGISTRY.get(name)(cfg, input_shape)
    if not callable(getattr(model, "forward_features", None)):
        raise ValueError("Only SEM_SEG_HEADS with forward_features method can be used as pixel decoder. "
                         f"Please implement forward_features for {name} to only return mask features.")
    return model


def build_transformer_decoder(cfg, in_channels, mask_classification=True):
    """mask2former/modeling/transformer_decoder/maskformer_transformer_decoder.py:22-27"""
    name = cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME
    return TRANSFORMER_DECODER_REGISTRY.get(name)(cfg, in_channels, mask_classification)


@SEM_SEG_HEADS_REGISTRY.register()
class MaskFormerHead(nn.Module):
    _version = 2

    @configurable
    def __init__(self, input_shape: Dict[str, ShapeSpec], *, num_classes: int, pixel_decoder: nn.Module,
                 pixel_decoder_name: str, loss_weight: float = 1.0, ignore_value: int = -1,
                 transformer_predictor: nn.Module, transformer_in_feature: str, frozen_pixel_decoder: bool = False,
                 frozen_mask_convs: bool = False, frozen_predictor: bool = False):
        super().__init__()
        input_shape = sorted(input_shape.items(), key=lambda x: x[1].stride)
        self.in_features = [k for k, v in input_shape]
        self.ignore_value = ignore_value
        self.common_stride = 4
        self.loss_weight = loss_weight
        self.pixel_decoder = pixel_decoder
        self.pixel_decoder_name = pixel_decoder_name
        self.predictor = transformer_predictor
        self.transformer_in_feature = transformer_in_feature
        self.num_classes = num_classes

    @classmethod
    def from_config(cls, cfg, input_shape: Dict[str, ShapeSpec]):
        tif = cfg.MODEL.MASK_FORMER.TRANSFORMER_IN_FEATURE
        if tif in ("transformer_encoder", "multi_scale_pixel_decoder"):
            in_ch = cfg.MODEL.SEM_SEG_HEAD.CONVS_DIM
        elif tif == "pixel_embedding":
            in_ch = cfg.MODEL.SEM_SEG_HEAD.MASK_DIM
        else:
            in_ch = input_shape[tif].channels
        return {
            "input_shape": {k: v for k, v in input_shape.items() if k in cfg.MODEL.SEM_SEG_HEAD.IN_FEATURES},
            "ignore_value": cfg.MODEL.SEM_SEG_HEAD.IGNORE_VALUE,
            "num_classes": cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES,
            "pixel_decoder": build_pixel_decoder(cfg, input_shape),
            "pixel_decoder_name": cfg.MODEL.SEM_SEG_HEAD.PIXEL_DECODER_NAME,
            "loss_weight": cfg.MODEL.SEM_SEG_HEAD.LOSS_WEIGHT,
            "transformer_in_feature": tif,
            "transformer_predictor": build_transformer_decoder(cfg, in_ch, mask_classification=True),
            "frozen_pixel_decoder": cfg.MODEL.SEM_SEG_HEAD.FROZEN_PIXEL_DECODER,
            "frozen_mask_convs": cfg.MODEL.SEM_SEG_HEAD.FROZEN_MASK_CONVS,
            "frozen_predictor": cfg.MODEL.SEM_SEG_HEAD.FROZEN_PREDICTOR,
        }

    def forward(self, features, mask=None, targets=None):
        return self.layers(features, mask, targets)

    def prefetch_prompts(self, targets, num_frames):
        """Optional, for callers that run the backbone and the head back to back on one clip: called BEFORE the backbone is
        enqueued, the annotation-only part of the visual-prompt sampler (and its one host round trip) overlaps the backbone
        instead of the pixel decoder.  `forward` starts it itself when the caller did not."""
        if targets is not None and hasattr(self.predictor, "prefetch_prompts"):
            self.predictor.prefetch_prompts(targets, num_frames)

    def layers(self, features, mask=None, targets=None):
        if self.pixel_decoder_name != "MSDeformAttnPixelDecoder":
            raise ValueError(f"pixel decoder {self.pixel_decoder_name} is outside the hot path (SURVEY.md section 2)")
        if targets is not None and hasattr(self.predictor, "prefetch_prompts"):
            # the annotation-only part of the visual-prompt sampler, on a side stream, ahead of the pixel decoder
            self.predictor.prefetch_prompts(targets, next(iter(features.values())).shape[0])
        mask_features, mask_features_bfe_conv, enc_features, multi_scale_features = \
            self.pixel_decoder.forward_features(features)
        if self.transformer_in_feature != "multi_scale_pixel_decoder":
            raise ValueError("only TRANSFORMER_IN_FEATURE='multi_scale_pixel_decoder' is on the hot path")
        return self.predictor(multi_scale_features, mask_features, mask_features_bfe_conv, mask, targets)
