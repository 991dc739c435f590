"""Builds the per-video `targets` dictionaries the clip loops and the prompt-as-query decoder consume (inference).

Counterpart of the inference branch of the reference's `PrepareTargets` (univs/prepare_targets.py):
    process_inference        :46-95     one dict per video: task, dataset, prompt type, sizes, file names, (palette,
                                        per-frame annotations for 'sot'); custom text prompts turn a video into a
                                        'grounding' request (:52-56)
    preprocess_text_prompt   :260-385   (is_train=False) 'grounding': expressions -> CLIP word / sentence features through
                                        the text prompt encoder; 'detection': the slice of the class-embedding table that
                                        belongs to the dataset (the whole table for raw videos and unknown datasets)
Training-time target preparation (`process`, :97-258, and the is_train branches) is out of scope of the inference path.
"""
from typing import List, Optional

import torch

from .modeling.transformer_decoder.univs_decoder import combined_datasets_category_info


def is_semseg_dataset(dataset_name: str) -> bool:
    return dataset_name.startswith("vspw")


class PrepareTargets:
    def __init__(self, num_frames: int = 1, max_num_masks: int = 30, text_prompt_enable: bool = False,
                 boxvis_enabled: bool = False, clip_class_embed_path="", thing_only_enabled: bool = False,
                 semantic_on: bool = False, custom_videos_text: Optional[List] = None):
        self.num_frames = num_frames
        self.max_num_masks = max_num_masks
        self.text_prompt_enable = text_prompt_enable
        self.boxvis_enabled = boxvis_enabled
        # class names as CLIP text embeddings, generated offline (a tensor may be passed instead of a path)
        self.clip_cls_text_emb = (clip_class_embed_path if isinstance(clip_class_embed_path, torch.Tensor)
                                  else torch.load(clip_class_embed_path, map_location="cpu"))
        self.thing_only_enabled = thing_only_enabled
        self.semantic_on = semantic_on
        custom_videos_text = custom_videos_text or []
        assert len(custom_videos_text) <= 1, "Only support a single video now"
        self.custom_videos_text = custom_videos_text

    def process_inference(self, targets, inter_image_size, device, text_prompt_encoder=None, image_size=None):
        """targets: the mapper's per-video dicts (`batched_inputs`); inter_image_size: the padded input size."""
        if len(self.custom_videos_text) > 0:
            assert len(self.custom_videos_text) == len(targets)
            for tv, expressions in zip(targets, self.custom_videos_text):
                tv["task"] = "grounding"
                tv["expressions"] = expressions
                tv["exp_obj_ids"] = list(range(len(expressions)))

        task = targets[0]["task"]
        if task == "grounding":
            prompt_type = "text"
        elif task == "detection":
            prompt_type = "text" if is_semseg_dataset(targets[0]["dataset_name"]) or self.semantic_on else "visual"
        else:
            prompt_type = "visual"          # 'sot'

        clip_gt_instances = []
        for tv in targets:
            out = {"video_len": tv["video_len"], "dataset_name": tv["dataset_name"], "task": tv["task"],
                   "num_frames": self.num_frames, "inter_image_size": inter_image_size, "image_size": image_size,
                   "file_names": tv["file_names"]}
            if "video_id" in tv:
                out["video_id"] = tv["video_id"],            # the reference stores a 1-tuple here (:80, trailing comma)
            tv["prompt_type"] = prompt_type
            out["prompt_type"] = prompt_type
            if "mask_palette" in tv:
                out["mask_palette"] = tv["mask_palette"]
            if task == "sot":
                out["instances"] = tv["instances"]
            clip_gt_instances.append(out)

        if task in {"detection", "grounding"}:
            self.preprocess_text_prompt(text_prompt_encoder, targets, clip_gt_instances, device=device, is_train=False)
        return clip_gt_instances

    def process(self, targets, images, device, text_prompt_encoder=None, is_train=True):
        raise NotImplementedError("training-time target preparation is out of scope of the inference hot path")

    def preprocess_text_prompt(self, text_prompt_encoder, targets, clip_gt_instances, valid_bool_clips=None, device="cpu",
                               num_max_instances=30, is_train=True):
        if is_train:
            raise NotImplementedError("training-time target preparation is out of scope of the inference hot path")
        for out, tv in zip(clip_gt_instances, targets):
            if tv["task"] == "grounding":
                if not ("expressions" in tv and len(tv["expressions"]) > 0):
                    continue                                 # nothing to ground: the dict stays without text prompts
                expressions, exp_obj_ids = tv["expressions"], tv["exp_obj_ids"]
                assert len(expressions) == len(exp_obj_ids), \
                    f"Mismatch number between expressions and exp_ids: {len(expressions)} and {len(exp_obj_ids)}"
                out["expressions"] = expressions
                out["exp_obj_ids"] = exp_obj_ids
                word, sentence, n_words = text_prompt_encoder.get_expression_prompt(expressions, device)
                out["exp_word_len"] = n_words[:num_max_instances]
                out["exp_word_feats"] = word[:num_max_instances]
                out["exp_sentence_feats"] = sentence[:num_max_instances]
                out["prompt_obj_ids"] = exp_obj_ids[:num_max_instances]
            elif tv["task"] == "detection":
                name = tv["dataset_name"]
                if name in {"flickr"}:
                    raise NotImplementedError("phrase grounding on Flickr (class names from phrases) is not built")
                if tv.get("is_raw_video", False) or name not in combined_datasets_category_info:
                    out["clip_cls_text_emb"] = self.clip_cls_text_emb
                else:
                    num_classes, start = combined_datasets_category_info[name]
                    out["clip_cls_text_emb"] = self.clip_cls_text_emb[start:start + num_classes]
