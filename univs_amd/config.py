"""Config surface of the hot path: a small attribute-dict `CfgNode` with yaml `_BASE_` inheritance and
`KEY VALUE` command-line overrides (yacs / detectron2 are absent on both boxes).

Only the keys that parameterise the hot path carry defaults here (SURVEY.md section 5 "Config / flags");
they restate the values of the reference's config adders:
  mask2former/config.py:6-129 (MASK_FORMER, SEM_SEG_HEAD, SWIN), univs/config.py:4-160 (INPUT, UniVS,
  BoxVIS.TEST) and detectron2's RESNETS/BACKBONE defaults.  Unknown keys found in a yaml file are
  accepted (the reference's recipes carry solver / dataset groups that are out of scope here).
"""
import ast
import copy
import os

import yaml


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    # ---- construction ---------------------------------------------------------------------------
    @staticmethod
    def _convert(d):
        if isinstance(d, dict) and not isinstance(d, CfgNode):
            n = CfgNode()
            for k, v in d.items():
                n[k] = CfgNode._convert(v)
            return n
        if isinstance(d, str):
            # yacs semantics: strings that are python literals ("(600, 1024)") become those literals
            try:
                lit = ast.literal_eval(d)
                if isinstance(lit, (tuple, list, int, float, bool)) or lit is None:
                    return lit
            except (ValueError, SyntaxError):
                pass
        return d

    def merge_from_other(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge_from_other(v)
            else:
                self[k] = CfgNode._convert(v)

    @staticmethod
    def _load_yaml_with_base(path):
        with open(path) as f:
            cfg = yaml.safe_load(f) or {}
        base = cfg.pop("_BASE_", None)
        if base is not None:
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(path), base)
            merged = CfgNode._load_yaml_with_base(base)
            CfgNode._convert(merged)
            node = CfgNode._convert(merged)
            node.merge_from_other(CfgNode._convert(cfg))
            return node
        return CfgNode._convert(cfg)

    def merge_from_file(self, path):
        self.merge_from_other(CfgNode._load_yaml_with_base(path))

    def merge_from_list(self, opts):
        """`["MODEL.SWIN.EMBED_DIM", "128", ...]` as on the reference's command line (train_net.py:362-363)."""
        assert len(opts) % 2 == 0, "override list must be KEY VALUE pairs"
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    node[p] = CfgNode()
                node = node[p]
            node[parts[-1]] = CfgNode._convert(val) if isinstance(val, str) else val


CN = CfgNode


def get_cfg():
    c = CN()
    c.MODEL = CN()
    m = c.MODEL
    m.META_ARCHITECTURE = "UniVS_Prompt"
    m.PIXEL_MEAN = [123.675, 116.280, 103.530]
    m.PIXEL_STD = [58.395, 57.120, 57.375]
    m.WEIGHTS = ""
    m.BACKBONE = CN(NAME="D2SwinTransformer", FREEZE_AT=0)
    m.RESNETS = CN(DEPTH=50, STEM_OUT_CHANNELS=64, STRIDE_IN_1X1=False, NORM="FrozenBN", NUM_GROUPS=1,
                   WIDTH_PER_GROUP=64, RES2_OUT_CHANNELS=256, RES5_DILATION=1,
                   OUT_FEATURES=["res2", "res3", "res4", "res5"])
    m.SWIN = CN(PRETRAIN_IMG_SIZE=224, PATCH_SIZE=4, EMBED_DIM=96, DEPTHS=[2, 2, 6, 2],
                NUM_HEADS=[3, 6, 12, 24], WINDOW_SIZE=7, MLP_RATIO=4.0, QKV_BIAS=True, QK_SCALE=None,
                DROP_RATE=0.0, ATTN_DROP_RATE=0.0, DROP_PATH_RATE=0.3, APE=False, PATCH_NORM=True,
                OUT_FEATURES=["res2", "res3", "res4", "res5"], USE_CHECKPOINT=False)
    m.SEM_SEG_HEAD = CN(NAME="MaskFormerHead", IGNORE_VALUE=255, NUM_CLASSES=133, LOSS_WEIGHT=1.0,
                        CONVS_DIM=256, MASK_DIM=256, NORM="GN", LANG_DIM=640,
                        PIXEL_DECODER_NAME="MSDeformAttnPixelDecoder",
                        IN_FEATURES=["res2", "res3", "res4", "res5"],
                        DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES=["res3", "res4", "res5"],
                        DEFORMABLE_TRANSFORMER_ENCODER_N_POINTS=4, DEFORMABLE_TRANSFORMER_ENCODER_N_HEADS=8,
                        COMMON_STRIDE=4, TRANSFORMER_ENC_LAYERS=6, FROZEN_PIXEL_DECODER=False,
                        FROZEN_MASK_CONVS=False, FROZEN_PREDICTOR=False)
    m.MASK_FORMER = CN(NHEADS=8, DROPOUT=0.1, DIM_FEEDFORWARD=2048, ENC_LAYERS=0, DEC_LAYERS=10,
                       PRE_NORM=False, HIDDEN_DIM=256, NUM_OBJECT_QUERIES=200,
                       TRANSFORMER_IN_FEATURE="multi_scale_pixel_decoder", ENFORCE_INPUT_PROJ=False,
                       SIZE_DIVISIBILITY=32,
                       TRANSFORMER_DECODER_NAME="VideoMultiScaleMaskedTransformerDecoderUniVS",
                       # mask2former/config.py:60-65, univs/config.py:90-91
                       TEST=CN(SEMANTIC_ON=True, INSTANCE_ON=False, PANOPTIC_ON=False, OBJECT_MASK_THRESHOLD=0.0,
                               OVERLAP_THRESHOLD=0.0, STABILITY_SCORE_THRESH=0.0, OVERLAP_THRESHOLD_ENTITY=0.5))
    m.BoxVIS = CN(TEST=CN(NUM_FRAMES=3, NUM_FRAMES_WINDOW=5, NUM_MAX_INST=50, CLIP_STRIDE=1,
                          LSJ_AUG_ENABLED=True, APPLY_CLS_THRES=0.05, MULTI_CLS_ON=True))   # univs/config.py:101-113
    # text tower of CLIP RN50x4 (TextEncoder.py:152-174 reads these; the reference sets them in its yaml recipes)
    m.CLIP = CN(RESNETS_DEPTH=200, BACKBONE_FREEZE_AT=0, WEIGHTS="")
    m.UniVS = CN(PROMPT_TYPE="category",
                 CLIP_CLASS_EMBED_PATH="datasets/concept_emb/combined_datasets_cls_emb_rn50x4.pth",
                 NUM_POS_QUERIES=30, VISUAL_PROMPT_ENCODER=True, TEXT_PROMPT_ENCODER=True,
                 LANGUAGE_ENCODER_ENABLE=True, PROMPT_AS_QUERIES=True, VISUAL_PROMPT_TO_IMAGE_ENABLE=True,
                 TEXT_PROMPT_TO_IMAGE_ENABLE=True, MASKDEC_ATTN_ORDER="casa",
                 MASKDEC_SELF_ATTN_MASK_TYPE="sep", DISABLE_LEARNABLE_QUERIES_SA1B=False,
                 VISUAL_PROMPT_PIXELS_PER_IMAGE=32, PROMPT_SELF_ATTN_LAYERS=-1,
                 POSITION_EMBEDDING_SINE3D="ArbitraryT")
    m.UniVS.TEST = CN(VIDEO_UNIFIED_INFERENCE_ENABLE=False, CLIP_STRIDE=1, NUM_PREV_FRAMES_MEMORY=5,
                      ENABLED_PREV_FRAMES_MEMORY=True, ENABLED_PREV_VISUAL_PROMPTS_FOR_GROUNDING=False,
                      DETECT_NEWLY_INTERVAL_FRAMES=1,
                      # univs/config.py:138-153
                      VIDEO_UNIFIED_INFERENCE_QUERIES="prompt", VIDEO_UNIFIED_INFERENCE_ENTITIES="",
                      BOX_NMS_THRESH=0.75, TEMPORAL_CONSISTENCY_THRESHOLD=0.05, DETECT_NEWLY_OBJECT_THRESHOLD=0.05,
                      CUSTOM_VIDEOS_ENABLE=False, CUSTOM_VIDEOS_TEXT=[],
                      SEMANTIC_EXTRACTION=CN(ENABLE=False))
    c.INPUT = CN(FORMAT="RGB", SAMPLING_FRAME_NUM=2, MIN_SIZE_TEST=800, MAX_SIZE_TEST=1333,
                 LSJ_AUG=CN(ENABLED=True, SQUARE_ENABLED=True, IMAGE_SIZE=1024, MIN_SCALE=0.25, MAX_SCALE=4.0))
    c.TEST = CN(DETECTIONS_PER_IMAGE=100)
    return c


def load_cfg(config_file=None, opts=None):
    cfg = get_cfg()
    if config_file:
        cfg.merge_from_file(config_file)
    if opts:
        cfg.merge_from_list(list(opts))
    return cfg
