"""Inference shell of the reference's `UniVS_Prompt` META_ARCH (univs/univs_prompt.py): the object the clip loops call
back into (`model.backbone`, `model.sem_seg_head`, `model.prepare_targets`, `model.text_prompt_encoder`) and its
`forward_inference` dispatch (:416-452):

    task 'grounding' / 'sot', or custom text prompts            -> InferenceVideoVOS.eval
    category-specified tasks (ytvis / ovis / vipseg / vspw,
    or custom videos) with unified inference enabled             -> InferenceVideoEntity.eval

The other branches of the reference (per-image COCO / ADE20k evaluation, the MinVIS / MDQE trackers of the non-unified
mode, EMA teacher weights, semantic-feature extraction, and all of training: losses, matcher, `forward` in train mode)
are out of scope of the hot path and raise.
"""
import torch
from torch import nn

from ...inference.video_entity import InferenceVideoEntity
from ...inference.video_vos import InferenceVideoVOS
from ...prepare_targets import PrepareTargets
from ...registry import META_ARCH_REGISTRY, configurable
from ..prompt_encoder import TextPromptEncoder


@META_ARCH_REGISTRY.register()
class UniVS_Prompt(nn.Module):
    @configurable
    def __init__(self, *, backbone, sem_seg_head, prepare_targets, text_prompt_encoder, inference_video_entity,
                 inference_video_vos, pixel_mean, pixel_std, video_unified_inference_enable: bool,
                 custom_videos_enable: bool, custom_videos_text):
        super().__init__()
        self.backbone = backbone
        self.sem_seg_head = sem_seg_head
        self.prepare_targets = prepare_targets
        self.text_prompt_encoder = text_prompt_encoder
        self.inference_video_entity = inference_video_entity
        self.inference_video_vos = inference_video_vos
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean, dtype=torch.float32).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std, dtype=torch.float32).view(-1, 1, 1), False)
        self.video_unified_inference_enable = video_unified_inference_enable
        self.custom_videos_enable = custom_videos_enable
        self.custom_videos_text = custom_videos_text

    @classmethod
    def from_config(cls, cfg):
        from ..build import build_backbone, build_sem_seg_head
        backbone = build_backbone(cfg)
        sem_seg_head = build_sem_seg_head(cfg, backbone.output_shape())
        lang_encoder = None
        if cfg.MODEL.UniVS.LANGUAGE_ENCODER_ENABLE and cfg.MODEL.UniVS.TEXT_PROMPT_ENCODER and cfg.MODEL.CLIP.WEIGHTS:
            from ..language import build_clip_language_encoder
            lang_encoder = build_clip_language_encoder(cfg)
        text_prompt_encoder = None
        if cfg.MODEL.UniVS.TEXT_PROMPT_ENCODER:
            text_prompt_encoder = TextPromptEncoder(lang_encoder=lang_encoder, num_frames=cfg.INPUT.SAMPLING_FRAME_NUM)
        test = cfg.MODEL.UniVS.TEST
        prepare_targets = PrepareTargets(num_frames=cfg.INPUT.SAMPLING_FRAME_NUM, max_num_masks=cfg.MODEL.UniVS.NUM_POS_QUERIES,
                                         text_prompt_enable=cfg.MODEL.UniVS.TEXT_PROMPT_ENCODER,
                                         clip_class_embed_path=cfg.MODEL.UniVS.CLIP_CLASS_EMBED_PATH,
                                         custom_videos_text=test.CUSTOM_VIDEOS_TEXT)
        return {
            "backbone": backbone, "sem_seg_head": sem_seg_head, "prepare_targets": prepare_targets,
            "text_prompt_encoder": text_prompt_encoder, "inference_video_entity": InferenceVideoEntity(cfg),
            "inference_video_vos": InferenceVideoVOS(cfg), "pixel_mean": cfg.MODEL.PIXEL_MEAN,
            "pixel_std": cfg.MODEL.PIXEL_STD, "video_unified_inference_enable": test.VIDEO_UNIFIED_INFERENCE_ENABLE,
            "custom_videos_enable": test.CUSTOM_VIDEOS_ENABLE, "custom_videos_text": test.CUSTOM_VIDEOS_TEXT,
        }

    @property
    def device(self):
        return self.pixel_mean.device

    def forward(self, batched_inputs):
        if self.training:
            raise NotImplementedError("training (losses, matcher) is out of scope of the inference hot path")
        return self.forward_inference(batched_inputs)

    @torch.no_grad()
    def forward_inference(self, batched_inputs):
        name = batched_inputs[0]["dataset_name"]
        if name.startswith("coco") or name.startswith("ade20k"):
            raise NotImplementedError("per-image evaluation (InferenceImageGenericSegmentation) is not built")
        if batched_inputs[0]["task"] in {"grounding", "sot"} or len(self.custom_videos_text):
            return self.inference_video_vos.eval(self, batched_inputs)          # prompt-specified tasks
        if self.video_unified_inference_enable or self.custom_videos_enable:
            if name.startswith(("ytvis", "ovis", "vipseg", "vspw")) or self.custom_videos_enable:
                return self.inference_video_entity.eval(self, batched_inputs)   # category-specified tasks
            raise ValueError(f"Not support to eval the dataset {name} yet")
        raise NotImplementedError("the non-unified trackers (MinVIS / MDQE style association) are not built: set "
                                  "MODEL.UniVS.TEST.VIDEO_UNIFIED_INFERENCE_ENABLE True")


@META_ARCH_REGISTRY.register()
class UniVS_Prompt_LongVideo(UniVS_Prompt):
    """`univs/univs_prompt_longvideo.py`: the long-video recipe differs from `UniVS_Prompt` in TRAINING only (several clips
    of one video per step, inter-clip re-identification loss, :347-438 / :469-589); its inference dispatch (:440-467) sends
    category-specified videos to the same unified entity loop and 'sot*' datasets to the VOS loop."""

    @torch.no_grad()
    def forward_inference(self, batched_inputs):
        name = batched_inputs[0]["dataset_name"]
        if name.startswith("coco") or name.startswith("ade20k"):
            raise ValueError(f"Not support to eval the image datasets {name} here")
        if self.video_unified_inference_enable:
            if name.startswith(("ytvis", "ovis", "vipseg", "vpsw")):      # ('vpsw': the reference's spelling, :451)
                return self.inference_video_entity.eval(self, batched_inputs)
            raise ValueError(f"Not support to eval the dataset {name} yet")
        if name.startswith("sot"):
            return self.inference_video_vos.eval(self, batched_inputs)
        raise NotImplementedError("the non-unified trackers (MinVIS / MDQE style association) are not built: set "
                                  "MODEL.UniVS.TEST.VIDEO_UNIFIED_INFERENCE_ENABLE True")


@META_ARCH_REGISTRY.register()
class MaskFormer_Video(nn.Module):
    """Inference consumer with the call pattern of `mask2former_video/video_maskformer_model.py:203-209` -- the second
    META_ARCHITECTURE the north star names: normalise + pad the clip's frames, `features = self.backbone(images.tensor)`,
    `outputs = self.sem_seg_head(features)` (NO targets).  Returns the head's raw outputs for the clip; the vanilla
    Mask2Former-video post-processing (`inference_video`: top-k over queries x classes, :281-330) and its own
    transformer decoder are a different model family, outside SURVEY.md 8a."""

    @configurable
    def __init__(self, *, backbone, sem_seg_head, num_frames, size_divisibility, pixel_mean, pixel_std,
                 dataset_name="ytvis_2021_dev"):
        super().__init__()
        self.backbone = backbone
        self.sem_seg_head = sem_seg_head
        # this meta-architecture calls the head without targets: it names the class vocabulary of the category-specified
        # first clip the decoder then builds (the decoder raises for a target-less call otherwise)
        pred = getattr(sem_seg_head, "predictor", None)
        if pred is not None and hasattr(pred, "default_dataset_name"):
            pred.default_dataset_name = dataset_name
        self.num_frames = num_frames
        self.size_divisibility = size_divisibility if size_divisibility >= 0 else getattr(backbone, "size_divisibility", 32)
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean, dtype=torch.float32).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std, dtype=torch.float32).view(-1, 1, 1), False)

    @classmethod
    def from_config(cls, cfg):
        from ..build import build_backbone, build_sem_seg_head
        backbone = build_backbone(cfg)
        return {"backbone": backbone, "sem_seg_head": build_sem_seg_head(cfg, backbone.output_shape()),
                "num_frames": cfg.INPUT.SAMPLING_FRAME_NUM, "size_divisibility": cfg.MODEL.MASK_FORMER.SIZE_DIVISIBILITY,
                "pixel_mean": cfg.MODEL.PIXEL_MEAN, "pixel_std": cfg.MODEL.PIXEL_STD}

    @property
    def device(self):
        return self.pixel_mean.device

    @torch.no_grad()
    def forward(self, batched_inputs):
        if self.training:
            raise NotImplementedError("training (losses, matcher) is out of scope of the inference hot path")
        from ...inference.video_entity import ImageList
        frames = [f.to(self.device) for video in batched_inputs for f in video["image"]]
        images = ImageList.from_tensors([(f - self.pixel_mean) / self.pixel_std for f in frames], self.size_divisibility)
        features = self.backbone(images.tensor)
        return self.sem_seg_head(features)
