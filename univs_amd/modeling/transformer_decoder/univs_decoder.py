"""Prompt-as-query masked transformer decoder of UniVS for the MI355X hot path (inference).

Interface, registration name and state-dict layout: the reference's
`VideoMultiScaleMaskedTransformerDecoderUniVS`
(univs/modeling/transformer_decoder/video_mask2former_transformer_decoder_univs.py:27-303; forward
:305-454, ProCA :456-496, prediction heads :498-567, prompt encoders :599-758, lang->vision :760-793,
memory-pool read :795-822, self-attention mask :824-848).

What is organised differently (same results up to fp32 re-association):
  * mask decode = HIP MFMA kernel (`ops.mask_decode`) writing the [Q,T,H,W] layout callers consume;
  * the 10 per-layer attention masks come from the fused `ops.mask_decode_attn`: bilinear resampling
    commutes with the channel contraction, so the mask features are resampled ONCE per clip to the three
    level sizes and each head contracts at the target resolution, thresholds (logit < 0 <=> sigmoid < .5)
    and applies the all-masked-row reset of :390 in the epilogue.  Full-resolution logits are produced
    only for the last layer -- the 9 intermediate [Q,T,H,W] tensors are never consumed at inference
    (callers delete `aux_outputs`, univs/inference/inference_video_entity.py:317) -- unless
    `return_aux_outputs=True`;
  * the attention mask is kept as [T, Q, HW] and broadcast over the 8 heads instead of being repeated;
  * shape-only tensors (the spatial part of the 3-D sine embedding, the self-attention mask) are cached.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from ... import ops
from ...layers import Conv2d, fp32_region, to_device_async
from ...registry import TRANSFORMER_DECODER_REGISTRY, configurable
from ...switches import SWITCHES
from ..position_encoding import PositionEmbeddingSine3D, PositionEmbeddingSine3DArbitraryT
from ..prompt_encoder import VisualPromptSampler
from .transformer_layers import MLP, CrossAttentionLayer, FFNLayer, SelfAttentionLayer

# (num_classes, start index) of every dataset inside the [3938, 640] CLIP class-embedding table
# (datasets/concept_emb/combined_datasets_category_info.py:7-24)
combined_datasets_category_info = {
    "imagenet": (1000, 0), "lvis": (1203, 1000), "burst": (1203, 1000), "ytvis21": (40, 2203),
    "ovis": (25, 2243), "bdd_track": (8, 2268), "objects365": (365, 2276), "coco_panoptic": (133, 2641),
    "coco": (80, 2641), "ade20k": (150, 2774), "vipseg": (124, 2924), "vspw": (124, 2924),
    "viposeg": (124, 2924), "ytvis19": (40, 3048), "entityseg_instance": (206, 3088),
    "entityseg_panoptic": (644, 3294),
}


class _LazyList:
    """A list whose None entries are produced on first access (`makers[i]()`): the per-level position embeddings, which only
    the prompt encoder reads once the cross-attention keys come out of ops.decoder_memory."""

    def __init__(self, items, makers):
        self._items, self._makers = list(items), dict(makers)

    def _get(self, i):
        if self._items[i] is None:
            self._items[i] = self._makers[i]()
        return self._items[i]

    def __len__(self):
        return len(self._items)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._get(j) for j in range(*i.indices(len(self._items)))]
        return self._get(i if i >= 0 else len(self._items) + i)

    def __iter__(self):
        return (self._get(i) for i in range(len(self._items)))


@TRANSFORMER_DECODER_REGISTRY.register()
class VideoMultiScaleMaskedTransformerDecoderUniVS(nn.Module):
    _version = 2

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        version = local_metadata.get("version", None)
        if version is None or version < 2:  # old checkpoints: static_query -> query_feat (:32-53)
            for k in list(state_dict.keys()):
                if k.startswith(prefix) and "static_query" in k:
                    state_dict[k.replace("static_query", "query_feat")] = state_dict.pop(k)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    @configurable
    def __init__(self, in_channels, mask_classification=True, *, num_classes: int, hidden_dim: int,
                 num_queries: int, nheads: int, dim_feedforward: int, dec_layers: int, pre_norm: bool,
                 mask_dim: int, enforce_input_project: bool, prompt_self_attn_layers: int = -1,
                 num_frames: int = 1, clip_class_embed_path, visual_prompt_sampler, num_dense_points: int,
                 text_prompt_enable: bool = True, prompt_as_queries: bool = True,
                 text_prompt_to_image_enable: bool = True, maskdec_self_attn_mask_type: str = "sep",
                 disable_learnable_queries_sa1b: bool = False, position_embedding_sin3d_type: str = "FixedT",
                 num_prev_frames_memory: int = 5, enabled_prev_frames_memory: bool = True,
                 enabled_prev_visual_prompts_for_grounding: bool = False,
                 semantic_extraction_enable: bool = False, return_aux_outputs: bool = False):
        super().__init__()
        assert mask_classification, "Only support mask classification model"
        self.mask_classification = mask_classification
        self.num_frames = num_frames
        N_steps = hidden_dim // 2
        self.position_embedding_sin3d_type = position_embedding_sin3d_type
        if position_embedding_sin3d_type == "FixedT":
            self.pe_layer = PositionEmbeddingSine3D(N_steps, normalize=True)
        else:
            assert position_embedding_sin3d_type == "ArbitraryT"
            self.pe_layer = PositionEmbeddingSine3DArbitraryT(N_steps, normalize=True)

        self.num_heads = nheads
        self.num_layers = dec_layers
        self.transformer_self_attention_layers = nn.ModuleList()
        self.transformer_cross_attention_layers = nn.ModuleList()
        self.transformer_ffn_layers = nn.ModuleList()
        self.transformer_prompt_self_attention_layers = nn.ModuleList()
        self.prompt_self_attn_layers = self.num_layers if prompt_self_attn_layers < 0 else prompt_self_attn_layers
        for i in range(self.num_layers):
            self.transformer_self_attention_layers.append(
                SelfAttentionLayer(hidden_dim, nheads, dropout=0.0, normalize_before=pre_norm))
            self.transformer_cross_attention_layers.append(
                CrossAttentionLayer(hidden_dim, nheads, dropout=0.0, normalize_before=pre_norm))
            self.transformer_ffn_layers.append(
                FFNLayer(hidden_dim, dim_feedforward, dropout=0.0, normalize_before=pre_norm))
            if i < self.prompt_self_attn_layers:
                self.transformer_prompt_self_attention_layers.append(CrossAttentionLayer(hidden_dim, nheads, dropout=0.0))
        self.decoder_norm = nn.LayerNorm(hidden_dim)
        self.num_queries = num_queries
        self.query_feat = nn.Embedding(num_queries, hidden_dim)
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.num_feature_levels = 3
        self.level_embed = nn.Embedding(self.num_feature_levels, hidden_dim)
        self.input_proj = nn.ModuleList()
        for _ in range(self.num_feature_levels):
            if in_channels != hidden_dim or enforce_input_project:
                self.input_proj.append(Conv2d(in_channels, hidden_dim, kernel_size=1))
            else:
                self.input_proj.append(nn.Sequential())
        self.mask_embed = MLP(hidden_dim, hidden_dim, mask_dim, 3)

        # CLIP class-embedding table [K, 640]: a tensor, or the path of a torch-saved tensor (:193)
        if isinstance(clip_class_embed_path, torch.Tensor):
            self.clip_cls_text_emb = clip_class_embed_path
        else:
            self.clip_cls_text_emb = torch.load(clip_class_embed_path, map_location="cpu")
        self.text_emb_dim = self.clip_cls_text_emb.shape[-1]
        self.vis2text_projection = nn.Linear(hidden_dim, self.text_emb_dim)
        self.text_norm = nn.LayerNorm(self.text_emb_dim)
        self.text2vis_projection = nn.Linear(self.text_emb_dim, hidden_dim)
        self.cls_temp = nn.Embedding(1, 1)
        self.reid_temp = nn.Embedding(1, 1)
        self.maskdec_self_attn_mask_type = maskdec_self_attn_mask_type
        self.prompt_detection = nn.Embedding(1, hidden_dim)
        self.prompt_sot = nn.Embedding(1, hidden_dim)
        self.prompt_grounding = nn.Embedding(1, hidden_dim)
        self.visual_prompt_sampler = visual_prompt_sampler
        self.num_dense_points = num_dense_points
        self.visual_prompt_enable = visual_prompt_sampler is not None
        self.text_prompt_enable = text_prompt_enable
        self.prompt_as_queries = prompt_as_queries
        self.text_prompt_to_image_enable = text_prompt_to_image_enable
        if text_prompt_to_image_enable:
            self.lang2vision_cross_attention_layer = CrossAttentionLayer(hidden_dim, nheads, dropout=0.0,
                                                                         need_weights=True)
        self.num_prev_frames_memory = max(num_prev_frames_memory, num_frames)
        self.enabled_prev_frames_memory = enabled_prev_frames_memory
        self.enabled_prev_visual_prompts_for_grounding = enabled_prev_visual_prompts_for_grounding
        self.semantic_extraction_enable = semantic_extraction_enable
        self.return_aux_outputs = return_aux_outputs
        self.frame_shard = None  # univs_amd.distributed.FrameShard: frames of the clip sharded over ranks
        # `MaskFormer_Video` calls the head WITHOUT targets; the meta-architecture opts in by naming the class vocabulary here.
        # None (default): a call without targets raises, as the reference's decoder does (it dereferences targets[0], :310)
        self.default_dataset_name = None
        self._clip_norm_cache = None
        self._proca_kin = None       # (feats, pos, feats + pos, feats contiguous) of the clip in flight: _proca_fused
        self._sa_mask_cache = {}
        with torch.no_grad():  # the reference's init for the two temperatures (:233-236)
            self.cls_temp.weight.fill_(math.log(1 / 0.07))
            self.reid_temp.weight.fill_(math.log(1 / 0.07))

    @classmethod
    def from_config(cls, cfg, in_channels, mask_classification):
        visual_prompt_sampler = None
        if cfg.MODEL.UniVS.VISUAL_PROMPT_ENCODER:
            visual_prompt_sampler = VisualPromptSampler(
                pretrain_img_size=cfg.INPUT.LSJ_AUG.IMAGE_SIZE, hidden_dim=cfg.MODEL.MASK_FORMER.HIDDEN_DIM,
                num_heads=cfg.MODEL.MASK_FORMER.NHEADS, num_frames=cfg.INPUT.SAMPLING_FRAME_NUM,
                num_prev_frames_memory=cfg.MODEL.UniVS.TEST.NUM_PREV_FRAMES_MEMORY,
                num_dense_points=cfg.MODEL.UniVS.VISUAL_PROMPT_PIXELS_PER_IMAGE,
                position_embedding_sin3d_type=cfg.MODEL.UniVS.POSITION_EMBEDDING_SINE3D,
                clip_stride=cfg.MODEL.BoxVIS.TEST.CLIP_STRIDE)
        ret = dict(in_channels=in_channels, mask_classification=mask_classification)
        ret["num_classes"] = cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES
        ret["hidden_dim"] = cfg.MODEL.MASK_FORMER.HIDDEN_DIM
        ret["num_queries"] = cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES
        ret["nheads"] = cfg.MODEL.MASK_FORMER.NHEADS
        ret["dim_feedforward"] = cfg.MODEL.MASK_FORMER.DIM_FEEDFORWARD
        assert cfg.MODEL.MASK_FORMER.DEC_LAYERS >= 1
        ret["dec_layers"] = cfg.MODEL.MASK_FORMER.DEC_LAYERS - 1  # :273-279
        ret["pre_norm"] = cfg.MODEL.MASK_FORMER.PRE_NORM
        ret["enforce_input_project"] = cfg.MODEL.MASK_FORMER.ENFORCE_INPUT_PROJ
        ret["mask_dim"] = cfg.MODEL.SEM_SEG_HEAD.MASK_DIM
        ret["num_frames"] = cfg.INPUT.SAMPLING_FRAME_NUM
        ret["clip_class_embed_path"] = cfg.MODEL.UniVS.CLIP_CLASS_EMBED_PATH
        ret["visual_prompt_sampler"] = visual_prompt_sampler
        ret["num_dense_points"] = cfg.MODEL.UniVS.VISUAL_PROMPT_PIXELS_PER_IMAGE
        ret["text_prompt_enable"] = cfg.MODEL.UniVS.TEXT_PROMPT_ENCODER
        ret["prompt_as_queries"] = cfg.MODEL.UniVS.PROMPT_AS_QUERIES
        ret["text_prompt_to_image_enable"] = cfg.MODEL.UniVS.TEXT_PROMPT_TO_IMAGE_ENABLE
        ret["maskdec_self_attn_mask_type"] = cfg.MODEL.UniVS.MASKDEC_SELF_ATTN_MASK_TYPE
        ret["disable_learnable_queries_sa1b"] = cfg.MODEL.UniVS.DISABLE_LEARNABLE_QUERIES_SA1B
        ret["prompt_self_attn_layers"] = cfg.MODEL.UniVS.PROMPT_SELF_ATTN_LAYERS
        ret["position_embedding_sin3d_type"] = cfg.MODEL.UniVS.POSITION_EMBEDDING_SINE3D
        ret["num_prev_frames_memory"] = cfg.MODEL.UniVS.TEST.NUM_PREV_FRAMES_MEMORY
        ret["enabled_prev_frames_memory"] = cfg.MODEL.UniVS.TEST.ENABLED_PREV_FRAMES_MEMORY
        ret["enabled_prev_visual_prompts_for_grounding"] = cfg.MODEL.UniVS.TEST.ENABLED_PREV_VISUAL_PROMPTS_FOR_GROUNDING
        ret["semantic_extraction_enable"] = cfg.MODEL.UniVS.TEST.SEMANTIC_EXTRACTION.ENABLE
        return ret

    # ---------------------------------------------------------------------------------------------
    @fp32_region
    def forward(self, x, mask_features, mask_features_bfe_conv=None, mask=None, targets=None):
        assert not self.training, "inference-only module (training is out of scope of the hot path)"
        self._proca_kin = None
        bt, c_m, h_m, w_m = mask_features.shape
        bs, t = 1, bt  # all input frames form one video at inference (:310)
        assert len(x) == self.num_feature_levels
        del mask
        dev = mask_features.device
        src, pos, size_list = [], [], []
        fs = self.frame_shard  # None, or this rank's view of a clip whose frames are sharded over ranks
        t_total = t if fs is None else fs.total(t)
        if targets is None:
            # `MaskFormer_Video`'s call, `self.sem_seg_head(features)` (mask2former_video/video_maskformer_model.py:209):
            # no caller-owned state -> a category-specified first clip (the reference's UniVS decoder dereferences
            # `targets[0]` unconditionally, :310-333, so this combination raises there)
            if self.default_dataset_name is None:
                raise ValueError("VideoMultiScaleMaskedTransformerDecoderUniVS.forward needs `targets` (task, dataset_name, "
                                 "prompt_type, frame indices); a meta-architecture that calls the head without them sets "
                                 "`predictor.default_dataset_name` first")
            targets = [{"task": "detection", "dataset_name": self.default_dataset_name, "prompt_type": "visual",
                        "num_frames": t_total, "first_frame_idx": 0, "frame_indices": torch.arange(t_total, device=dev)}]
        frame_indices_host = None                        # the clip loops hand over host indices: the temporal embedding is cached by value
        if "frame_indices" in targets[0]:
            fi = torch.stack([tv["frame_indices"] for tv in targets])
            if not fi.is_cuda:
                frame_indices_host = fi
            frame_indices = to_device_async(fi, dev)   # (no host stall on the stream)
        else:
            frame_indices = torch.arange(t_total, device=dev)[None].repeat(bs, 1)
        if fs is not None:
            assert frame_indices.shape[1] == t_total, "targets['frame_indices'] must list the frames of ALL ranks"
            self._frame_indices_all = frame_indices
            frame_indices = frame_indices[:, fs.local_slice(t)]
            if frame_indices_host is not None:
                frame_indices_host = frame_indices_host[:, fs.local_slice(t)]
        pos_t_clip = None                                # pos_t [b, t, 2F] of this clip, shared by the levels
        if self.position_embedding_sin3d_type != "FixedT" and hasattr(self.pe_layer, "temporal") and not torch.is_grad_enabled():
            pos_t_clip = self.pe_layer.temporal(frame_indices_host if frame_indices_host is not None else frame_indices, dev)
        mem_fused = [None] * self.num_feature_levels     # (memory, key) of a level from ONE kernel (ops.decoder_memory), where covered
        pos_makers = {}                                  # level -> closure that materialises its position embedding on demand (local:
                                                         # nothing of a forward pass outlives it on the module)
        for i in range(self.num_feature_levels):
            size_list.append(tuple(int(s) for s in x[i].shape[-2:]))
            xi = x[i].view(bs, t, -1, size_list[-1][0], size_list[-1][1])
            fixed_t = self.position_embedding_sin3d_type == "FixedT"

            def make_pos(xi=xi, fixed_t=fixed_t):          # [b,t,C,h,w] -> [hw, bt, C]
                p = self.pe_layer(xi) if fixed_t else self.pe_layer(xi, frame_indices, pos_t=pos_t_clip)
                return p.flatten(3).flatten(0, 1).permute(2, 0, 1)
            xin = self.input_proj[i](x[i])
            if xin.is_cuda and bs == 1 and not torch.is_grad_enabled():
                # NCHW -> [hw, t, C] + level embedding, and the same + position embedding, in one pass over the features; the
                # position embedding itself is only materialised if the prompt encoder asks for it
                yx, pz = self.pe_layer.forward_separable(xi) if fixed_t else self.pe_layer.forward_separable(xi, frame_indices, pos_t=pos_t_clip)
                mem_fused[i] = ops.decoder_memory(xin, self.level_embed.weight[i], yx, pz[0])
            if mem_fused[i] is not None:
                src.append(mem_fused[i][0])
                pos.append(None)
                pos_makers[i] = make_pos
            else:
                pos.append(make_pos())
                s = xin.flatten(2) + self.level_embed.weight[i][None, :, None]
                src.append(s.permute(2, 0, 1))
        if any(m is not None for m in mem_fused):
            pos = _LazyList(pos, pos_makers)

        query_embed = self.query_embed.weight.unsqueeze(1).repeat(1, bt, 1)
        output = self.query_feat.weight.unsqueeze(1).repeat(1, bt, 1)
        prompt_feats_dense = prompt_pe_dense = None
        if self.prompt_as_queries:
            prompt_feats, prompt_pe, prompt_feats_dense, prompt_pe_dense, _ = \
                self.forward_prompt_encoder(src, pos, size_list, targets, t)
            if prompt_feats is not None:
                output = torch.cat([output, prompt_feats])
                prompt_pe = prompt_pe if prompt_pe is not None else prompt_feats
                query_embed = torch.cat([query_embed, prompt_pe])
            output = self.forward_transformer_prompt_self_attention_layer(
                0, output, query_embed, prompt_feats_dense, prompt_pe_dense)
            query_embed = torch.cat([query_embed[:self.num_queries], output[self.num_queries:]])

        task = targets[0]["task"]
        # mask features resampled once per clip to the three attention-mask resolutions
        mf = mask_features.float().contiguous()
        feat_lowres = {}
        h_m2, w_m2 = mf.shape[-2] >> 1, mf.shape[-1] >> 1
        if mf.is_cuda and set(size_list) == {(h_m2, w_m2), (h_m2 >> 1, w_m2 >> 1), (h_m2 >> 2, w_m2 >> 2)}:
            pyr = ops.bilinear_pyramid3(mf)            # strides 8 / 16 / 32 against stride 4: one pass over the features
            if pyr is not None:
                feat_lowres = {tuple(int(v) for v in p_.shape[-2:]): p_ for p_ in pyr}
        for sz in set(size_list):
            if sz not in feat_lowres:
                feat_lowres[sz] = ops.bilinear_resample(mf, sz)

        predictions_class, predictions_mask, predictions_embds, predictions_reid = [], [], [], []
        want_full = self.return_aux_outputs

        def heads(out_tokens, target_size, last):
            cls_, msk_, attn_, reid_ = self.forward_prediction_heads(
                out_tokens, mf, feat_lowres[target_size], task, targets, t, need_masks=(last or want_full),
                t_total=t_total, need_class=(last or want_full), deferred_mask=SWITCHES.fused_cross_attention and not last)
            predictions_class.append(cls_)
            predictions_mask.append(msk_)
            predictions_embds.append(out_tokens.view(out_tokens.shape[0], bs, t, -1).permute(1, 0, 2, 3))
            predictions_reid.append(reid_)
            return attn_

        attn_mask = heads(output, size_list[0], self.num_layers == 0)
        num_queries_lp = output.shape[0]
        self_attn_mask = self.generate_self_attn_mask(bs, t_total, num_queries_lp, dev, targets[0]["dataset_name"], task)
        query_embed_all = query_embed if fs is None else fs.all_gather_frames(query_embed, dim=1)
        # cross-attention inputs per level, contiguous and built once (every third layer reuses them)
        mem = [m[0] if m is not None else s_.contiguous() for m, s_ in zip(mem_fused, src)]
        mem_key = [m[1] if m is not None else (src[i_] + pos[i_]).contiguous() for i_, m in enumerate(mem_fused)]
        # key / value projections of the cross-attention: the layers i, i + L, i + 2 L, ... attend to the same level with
        # different weights -- ONE Linear per level for all their keys (N = 256 x layers), one for their values: the level's
        # memory is read once instead of once per layer; a layer takes its 256-column slice in place
        kv_proj = self._cross_kv(mem, mem_key) if (mem[0].is_cuda and not torch.is_grad_enabled()
                                                   and not self.transformer_cross_attention_layers[0].need_weights) else None
        for i in range(self.num_layers):
            if self.prompt_as_queries and 0 < i < self.prompt_self_attn_layers:
                output = self.forward_transformer_prompt_self_attention_layer(
                    i, output, query_embed, prompt_feats_dense, prompt_pe_dense)
            lvl = i % self.num_feature_levels
            # per-frame masked cross-attention; attn_mask [T, Q', HW_l] (rows already reset per :390)
            output = self.transformer_cross_attention_layers[i](
                output, mem[lvl], memory_mask=attn_mask, pos=None, query_pos=query_embed, key=mem_key[lvl],
                kv=None if kv_proj is None else kv_proj[i])
            # spatio-temporal self-attention over Q'*T tokens: 'Q (B T) C -> (Q T) B C'
            Qn = output.shape[0]
            if fs is None:
                o = output.reshape(Qn * t, bs, -1)
                qe = query_embed.reshape(Qn * t, bs, -1)
                o = self.transformer_self_attention_layers[i](o, tgt_mask=self_attn_mask, query_pos=qe)
                output = o.reshape(Qn, bs * t, -1)
            else:
                # the one collective of the layer: every rank gets the query states of all frames ([Q', T_loc, C] -> [Q', T, C]);
                # each rank then evaluates the self-attention ONLY for the query rows of its own frames (Q' T_loc rows against
                # all Q' T keys): the term shrinks with the number of ranks instead of being repeated on each
                # (6.7 ms of a 179-ms 40-frame clip on one GPU; evaluated redundantly it capped 8 GPUs at ~6.2x).
                full = fs.all_gather_frames(output, dim=1)
                kv = full.reshape(Qn * t_total, bs, -1)
                kv_pos = query_embed_all.reshape(Qn * t_total, bs, -1)
                rows_mask = None
                if self_attn_mask is not None:
                    rows_mask = self._sa_mask_rows(self_attn_mask, Qn, t_total, fs.local_slice(t))
                o = self.transformer_self_attention_layers[i](output.reshape(Qn * t, bs, -1), tgt_mask=rows_mask,
                                                              query_pos=query_embed.reshape(Qn * t, bs, -1), kv=kv, kv_pos=kv_pos)
                output = o.reshape(Qn, t, -1)
            output = self.transformer_ffn_layers[i](output)
            attn_mask = heads(output, size_list[(i + 1) % self.num_feature_levels], i == self.num_layers - 1)

        embds_norm_last = self.decoder_norm(predictions_embds[-1])
        out = {
            "pred_logits": predictions_class[-1],
            "pred_masks": predictions_mask[-1],
            "aux_outputs": [],
            "pred_embds": embds_norm_last,
            "pred_reid_logits": predictions_reid[-1],
        }
        if self.return_aux_outputs:
            out["aux_outputs"] = [
                {"pred_logits": a, "pred_masks": b, "pred_reid_logits": c, "pred_embds": self.decoder_norm(d)}
                for a, b, c, d in zip(predictions_class[:-1], predictions_mask[:-1], predictions_reid[:-1],
                                      predictions_embds[:-1])]
        if self.semantic_extraction_enable:
            out.update({"pred_embds": predictions_embds[-1][0].permute(1, 2, 0),
                        "mask_features": mask_features.view(bs, t, c_m, h_m, w_m)[0]})
        return out

    # ---------------------------------------------------------------------------------------------
    def forward_transformer_prompt_self_attention_layer(self, i, output, query_emb, prompt_feats_dense,
                                                        prompt_pos_dense):
        """ProCA (:456-496): every prompt query cross-attends only to its own prompt tokens
        (batch = Q_p * T, query length 1, key length 1 + L)."""
        if output.shape[0] == self.num_queries:
            return output
        nq = self.num_queries
        output_learn, output_prompt = output[:nq], output[nq:]
        query_emb_prompt = query_emb[nq:]
        layer = self.transformer_prompt_self_attention_layers[i]
        if (SWITCHES.fused_proca and output.is_cuda and output.dtype == torch.float32 and prompt_pos_dense is not None
                and not layer.normalize_before and not layer.need_weights and not torch.is_grad_enabled()):
            o = self._proca_fused(layer, output_prompt, query_emb_prompt, prompt_feats_dense, prompt_pos_dense)
            if o is not None:
                return torch.cat([output_learn, o])
        mem = torch.cat([output_prompt.unsqueeze(1), prompt_feats_dense], dim=1)     # Q_p x (1+L) x T x C
        mem = mem.transpose(0, 1).flatten(1, 2)                                     # (1+L) x Q_pT x C
        if prompt_pos_dense is not None:
            mpos = torch.cat([query_emb_prompt.unsqueeze(1), prompt_pos_dense], dim=1).transpose(0, 1).flatten(1, 2)
            qpos = query_emb_prompt.flatten(0, 1)[None]
        else:
            mpos, qpos = None, None
        Q_p, NT, _ = output_prompt.shape
        o = self.transformer_prompt_self_attention_layers[i](output_prompt.flatten(0, 1)[None], mem,
                                                             pos=mpos, query_pos=qpos)
        return torch.cat([output_learn, o.view(Q_p, NT, -1)])

    def _proca_fused(self, layer, output_prompt, query_emb_prompt, feats_dense, pos_dense):
        """One ProCA layer (post-norm CrossAttentionLayer, transformer_layers.py:95-115) without building `memory`: the first key /
        value are the prompt query's own state (k0 from state + query position, v0 from the state: the same few-rows launch as q);
        the dense tokens' keys come from `feats + pos` -- the sum is made ONCE per clip, it does not change across the layers -- and
        their values from `feats`, two tall Linears on the tokens in the memory pool's layout [Q_p, L, T, C]; one attention launch
        (ops.proca_attention); out_proj + residual + LayerNorm in one few-rows launch.  None: not covered (the caller's path)."""
        from ...layers import linear, linear_norm
        mha = layer.multihead_attn
        E, h = mha.embed_dim, mha.num_heads
        Q_p, T, _ = output_prompt.shape
        L = feats_dense.shape[1]
        if E // h != 32 or tuple(feats_dense.shape) != (Q_p, L, T, E) or tuple(pos_dense.shape) != (Q_p, L, T, E):
            return None
        x = output_prompt.reshape(Q_p * T, E)
        qkv0 = ops.small_linear(x, mha.in_proj_weight, mha.in_proj_bias, x_add=query_emb_prompt.reshape(Q_p * T, E), add_features=2 * E)
        if qkv0 is None:
            return None
        c = self._proca_kin
        if c is None or c[0] is not feats_dense or c[1] is not pos_dense:
            # the dense tokens do not change across the layers: their keys / values for ALL ProCA layers come out of ONE Linear each
            # (9 E output features, stored layer by layer: a column-blocked epilogue), as the cross-attention's do per level (_cross_kv)
            kin, vin = (feats_dense + pos_dense).contiguous(), feats_dense.contiguous()
            kv_all = self._proca_kv_all(kin.view(-1, E), vin.view(-1, E), E)
            c = self._proca_kin = (feats_dense, pos_dense, kin, vin, kv_all)
        li = next((j for j, m in enumerate(self.transformer_prompt_self_attention_layers) if m is layer), None)
        if c[4] is not None and li is not None:
            kd, vd = c[4][0][li].view(Q_p, L, T, E), c[4][1][li].view(Q_p, L, T, E)
        else:
            wk, bk = mha._packed_rows(E, E)
            wv, bv = mha._packed_rows(2 * E, E)
            kd = linear(c[2].view(-1, E), wk, bk).view(Q_p, L, T, E)
            vd = linear(c[3].view(-1, E), wv, bv).view(Q_p, L, T, E)
        out = ops.proca_attention(qkv0, kd, vd, h)
        if out is None:
            return None
        return linear_norm(out, mha.out_proj, x, layer.norm).view(Q_p, T, E)

    def _proca_kv_all(self, kin, vin, E):
        """(K, V) of the dense prompt tokens for every ProCA layer: two [n_layers, rows, E] tensors from two Linears with n_layers E
        output features (the layers' key / value rows of `in_proj`, concatenated once per weight version); None when not covered."""
        mods = [layer.multihead_attn for layer in self.transformer_prompt_self_attention_layers]
        if any(m.embed_dim != E for m in mods) or not kin.is_cuda:
            return None
        key = tuple((m.in_proj_weight.data_ptr(), m.in_proj_weight._version, m.in_proj_bias._version) for m in mods) + (str(kin.device),)
        c = self.__dict__.get("_proca_kv_cache")
        if c is None or c[0] != key:
            with torch.no_grad():
                c = (key, torch.cat([m.in_proj_weight[E:2 * E] for m in mods]).contiguous(), torch.cat([m.in_proj_bias[E:2 * E] for m in mods]).contiguous(),
                     torch.cat([m.in_proj_weight[2 * E:] for m in mods]).contiguous(), torch.cat([m.in_proj_bias[2 * E:] for m in mods]).contiguous())
            self.__dict__["_proca_kv_cache"] = c
        rows = kin.shape[0]
        k_all = ops.linear_blocked(kin, c[1], c[2], rows, E)
        v_all = ops.linear_blocked(vin, c[3], c[4], rows, E) if k_all is not None else None
        if v_all is None:
            return None
        return k_all[0], v_all[0]                                    # [n_layers, rows, E]

    def _clip_normalized(self, like):
        c = self._clip_norm_cache
        if c is None or c.device != like.device or c.dtype != like.dtype:
            c = F.normalize(self.clip_cls_text_emb.to(like), p=2, dim=-1).detach()
            self._clip_norm_cache = c
        return c

    def forward_prediction_heads(self, output, mask_features, feat_lowres, task, targets, t, need_masks,
                                 t_total=None, need_class=True, deferred_mask=False):
        """:498-567.  output [Q', T, C] (batch 1); mask_features [T, C, H, W]; feat_lowres [T, C, h, w].
        Returns (class logits [1,Q',K], mask logits [1,Q',T,H,W] or None, attn mask bool [T,Q',hw], reid)."""
        bs = 1
        fs = self.frame_shard
        t_total = t if t_total is None else t_total

        def mean_over_frames(x):  # x [T_loc, ...] -> [1, ...]: mean over the clip's frames (all ranks)
            if fs is None:
                return x.view(bs, t, *x.shape[1:]).mean(1)
            return fs.all_reduce_sum(x.sum(0, keepdim=True)) / float(t_total)

        # decoder_norm(output) [Q', T, C] feeds the mask-embedding MLP and -- only where class logits / re-id scores are formed -- the
        # heads below: on the GPU norm + MLP are one launch (layers.MLP: ops.small_mlp), and the normalised rows are written out only
        # where somebody reads them
        need_normed = need_class or (self.prompt_as_queries and task == "grounding")
        me = self.mask_embed(output, transpose01=True, in_norm=self.decoder_norm, want_normed=need_normed)
        mask_embed_early, decoder_qt = me if need_normed else (me, None)
        decoder_output = decoder_qt.transpose(0, 1) if decoder_qt is not None else None   # [T, Q', C] (a view)
        # class logits of the intermediate layers are only ever returned as aux outputs (the attention mask of the next
        # layer depends on the mask embedding alone): the 8 launches of this head run where their result is used
        outputs_class = self.vis2text_projection(decoder_output) if need_class else None
        if not need_class:
            pass
        elif task != "grounding":
            clip = self._clip_normalized(outputs_class)
            outputs_class = F.normalize(outputs_class, p=2, dim=-1)
            outputs_class = torch.einsum("bqc,kc->bqk", outputs_class, clip)
            outputs_class = mean_over_frames(outputs_class)
            outputs_class = outputs_class * self.cls_temp.weight.exp()
        else:
            clip_exp = torch.stack([tv["exp_sentence_feats"][:, 0] for tv in targets]).to(outputs_class).detach()
            outputs_class = mean_over_frames(outputs_class)
            outputs_class = torch.einsum("bqc,bkc->bqk", outputs_class, clip_exp)

        mask_embed = mask_embed_early                               # [T, Q', C] (written that way by the last Linear's kernel, or a view)
        outputs_reid = [None] * bs
        if self.prompt_as_queries and task == "grounding":
            assert len(targets) == 1, "Only support bacth size is 1 now"
            nq = self.num_queries
            output_norm = F.normalize(decoder_output, p=2, dim=-1)
            outputs_reid = torch.einsum("BqC,BkC->Bqk", output_norm, output_norm[:, nq:])
            outputs_reid = mean_over_frames(outputs_reid)
            l4p_indices = outputs_reid[:, :nq].flatten(0, -2).argmax(0)  # [Q_p]
            # mask_p <- (mask_p + mask_{l4p}) / 2  (:547) is linear in the mask embedding
            mask_embed = torch.cat([mask_embed[:, :nq], (mask_embed[:, nq:] + mask_embed[:, l4p_indices]) * 0.5], 1)
        mask_embed = mask_embed.contiguous()

        outputs_mask = None
        if need_masks:
            outputs_mask = ops.mask_decode(mask_embed, mask_features).unsqueeze(0)  # [1, Q', T, H, W]
        # [T, Q', hw] bool; `deferred_mask` (the layer loop): an ops.DeferredMask -- the fully-masked-row rule is applied by the
        # cross-attention kernel that reads the mask, not by a pass of its own
        attn_mask = ops.mask_decode_attn(mask_embed, feat_lowres, deferred=deferred_mask)
        return outputs_class, outputs_mask, attn_mask, outputs_reid

    # ---------------------------------------------------------------------------------------------
    @fp32_region
    def prefetch_prompts(self, targets, num_frames):
        """Called by the head BEFORE it enqueues the pixel decoder: starts the annotation-only work of the visual-prompt
        sampler for this clip on a side stream (VisualPromptSampler.prefetch).  A no-op where `forward_prompt_encoder`
        would not sample visual prompts from `targets[0]`, for CPU tensors and for frame-sharded clips."""
        if (not self.prompt_as_queries or self.visual_prompt_sampler is None or self.frame_shard is not None
                or len(targets) != 1 or torch.is_grad_enabled()):
            return
        tv = targets[0]
        if "_prompt_prefetch" in tv:       # the caller already started it (before the backbone)
            return
        if tv.get("task") == "sot" or tv.get("prompt_type") == "visual":
            self.visual_prompt_sampler.prefetch(tv, num_frames)

    def forward_prompt_encoder(self, src, pos, size_list, targets, num_frames=None, prompt_type=None,
                               use_all_prev_frames=False):
        """:599-758 (inference branches)."""
        if num_frames is None:
            num_frames = self.num_frames
        device = src[0].device
        tasks = [tv["task"] for tv in targets]
        assert all(tk == tasks[0] for tk in tasks)
        prompt_feats_dense, prompt_pe_dense, l2v = None, None, None

        fs = self.frame_shard
        if tasks[0] == "sot" or targets[0]["prompt_type"] == "visual" or prompt_type == "visual":
            enc = self.visual_prompt_sampler.visual_prompt_encoder
            if fs is not None:
                # the sampler runs replicated over the whole clip: position tokens are recomputed for all frames, the
                # features of the other ranks' frames are zeros and `feature_reduce` sums the token features over ranks
                src, pos = self._prompt_level_of_whole_clip(src, pos, size_list, num_frames)
                enc.feature_reduce = fs.all_reduce_sum
            else:
                enc.feature_reduce = None
            prompt_tuple = self.visual_prompt_sampler.process_per_batch(
                src, pos, size_list, targets, False, use_all_prev_frames=use_all_prev_frames)
            prompt_pe_dense, prompt_feats_dense = prompt_tuple[:2]
            if prompt_feats_dense is None:
                return None, None, None, None, None
            # mean over non-blank tokens (blank = all-zero embedding), :640-650
            output_prompt = query_embed_prompt = None
            if SWITCHES.fused_sampler and prompt_feats_dense.is_cuda:     # two launches instead of sixteen (csrc/prompt_sampler.hip: ps_token_mean)
                output_prompt = ops.token_mean(prompt_feats_dense, self.prompt_sot.weight)
                query_embed_prompt = ops.token_mean(prompt_pe_dense) if output_prompt is not None else None
            if query_embed_prompt is None:
                nb_f = torch.logical_not((prompt_feats_dense == 0).all(dim=-1)).unsqueeze(-1).sum(1).clamp(min=1)
                nb_p = torch.logical_not((prompt_pe_dense == 0).all(dim=-1)).unsqueeze(-1).sum(1).clamp(min=1)
                prompt_feats_mean = prompt_feats_dense.sum(1) / nb_f
                prompt_pe_mean = prompt_pe_dense.sum(1) / nb_p
                query_embed_prompt = prompt_pe_mean
                output_prompt = prompt_feats_mean + self.prompt_sot.weight.view(1, 1, -1)
            if "prompt_feats" in targets[0]:
                assert len(targets) == 1, "Only support batch size is 1 now"
                prompt_pe_dense, prompt_feats_dense = self.extract_prompt_features_from_memoey_pool(
                    targets, prompt_pe_dense, prompt_feats_dense)
            if fs is not None:      # back to this rank's frames
                sl = fs.local_slice(num_frames)
                output_prompt, query_embed_prompt = output_prompt[:, sl], query_embed_prompt[:, sl]
                prompt_feats_dense, prompt_pe_dense = prompt_feats_dense[:, :, sl], prompt_pe_dense[:, :, sl]
            return output_prompt, query_embed_prompt, prompt_feats_dense, prompt_pe_dense, l2v

        if tasks[0] == "detection":
            batch = []
            for tv in targets:
                name = tv["dataset_name"]
                assert name in combined_datasets_category_info
                num_classes, start_idx = combined_datasets_category_info[name]
                emb = self.clip_cls_text_emb[start_idx:start_idx + num_classes].to(device)
                assert len(emb) == num_classes
                e = self.text2vis_projection(self.text_norm(emb))
                batch.append(e[:, None].repeat(1, num_frames, 1))
            feats = torch.stack(batch, dim=1).flatten(1, 2)  # num_classes x NT x C
            if self.text_prompt_to_image_enable:
                feats, l2v = self.forward_lang_to_vision(feats, src, size_list, num_frames, tasks[0])
            return (feats + self.prompt_detection.weight.view(1, 1, -1), feats, feats.unsqueeze(1), None, l2v)

        if tasks[0] == "grounding":
            batch = []
            for tv in targets:
                fsl = slice(0, num_frames) if self.frame_shard is None else self.frame_shard.local_slice(num_frames)
                w = tv["exp_word_feats"][..., fsl, :]       # num_exp x 77 x T x 640 (this rank's frames)
                s = tv["exp_sentence_feats"][..., fsl, :]   # num_exp x T x 640
                num_exps, len_sentence = w.shape[:2]
                ef = torch.cat([s[:, None], w], dim=1).flatten(0, 1).to(device)
                batch.append(self.text2vis_projection(self.text_norm(ef)))
            feats = torch.stack(batch, dim=1).flatten(1, 2)  # num_exp*(1+77) x NT x C
            if self.text_prompt_to_image_enable:
                feats, l2v = self.forward_lang_to_vision(feats, src, size_list, num_frames, tasks[0])
            prompt_feats_dense = feats.view(num_exps, len_sentence + 1, *feats.shape[1:])
            sentence = prompt_feats_dense[:, 0]
            output_prompt = sentence + self.prompt_grounding.weight.view(1, 1, -1)
            if "masks" in targets[0] and self.enabled_prev_visual_prompts_for_grounding:
                # NOTE the reference unpacks (pe, feats) as (feats, pe) here (:737-739); this path is
                # disabled in every shipped config (univs/config.py:148) and the swap is kept as is.
                vis_feats, vis_pe = self.visual_prompt_sampler.process_per_batch(
                    src, pos, size_list, targets, False, use_all_prev_frames=use_all_prev_frames)[:2]
                if vis_feats is not None:
                    prompt_feats_dense = torch.cat([vis_feats, prompt_feats_dense], dim=1)
            return output_prompt, sentence, prompt_feats_dense, None, l2v
        raise ValueError(tasks[0])

    def _prompt_level_of_whole_clip(self, src, pos, size_list, t_local):
        """Frame-sharded clip: the sampler's feature level as `[HW, T_total, C]` lists -- this rank's frames in place, zeros
        elsewhere -- and its position embedding for ALL frames (a function of shape and frame indices only)."""
        fs = self.frame_shard
        li = self.visual_prompt_sampler.prompt_feature_level_index
        h, w = size_list[li]
        t_total = fs.total(t_local)
        s_loc = src[li]
        s_full = s_loc.new_zeros(s_loc.shape[0], t_total, s_loc.shape[2])
        s_full[:, fs.local_slice(t_local)] = s_loc
        shape_only = s_loc.new_empty(1).expand(1, t_total, 1, h, w)
        if self.position_embedding_sin3d_type == "FixedT":
            p = self.pe_layer(shape_only)
        else:
            p = self.pe_layer(shape_only, self._frame_indices_all)
        p_full = p.flatten(3).flatten(0, 1).permute(2, 0, 1)
        src, pos = list(src), list(pos)
        src[li], pos[li] = s_full, p_full
        return src, pos

    def forward_lang_to_vision(self, prompt_feats, src, size_list, num_frames, task_type):
        """:760-793: one cross-attention of all text tokens against the concatenation of the 3 levels."""
        assert task_type in {"grounding", "detection"}
        src_flatten = torch.cat(src)
        feats, w = self.lang2vision_cross_attention_layer(prompt_feats, src_flatten)
        w = w / torch.max(w, dim=-1, keepdim=True)[0].clamp(min=1e-6)
        if task_type == "grounding":
            w = w.view(w.shape[0], -1, 78, w.shape[-1])[:, :, 0]
        w = torch.split(w, [s.shape[0] for s in src], dim=-1)
        w = [wi.reshape(-1, num_frames, wi.shape[1], h, ww).transpose(1, 2) for wi, (h, ww) in zip(w, size_list)]
        return feats, w

    @torch.no_grad()
    def extract_prompt_features_from_memoey_pool(self, targets, prompt_pe_dense, prompt_feats_dense):
        """:795-822: prompt tokens of the first-appearance frame + of the last `num_prev_frames_memory`
        frames of the pool, replicated over the clip's frames."""
        assert len(targets) == 1
        tv = targets[0]
        num_frames = prompt_feats_dense.shape[2]
        n_inst, _, e_idx = tv["prompt_feats"].shape[:3]
        first = tv["first_appear_frame_idxs"].clone()
        first[first >= e_idx - 1] = -1
        idx = torch.arange(n_inst, device=first.device)
        out = []
        for pool in (tv["prompt_feats"], tv["prompt_pe"]):
            f0 = pool[idx, :, first]                                         # n x R x C
            prev = pool[:, :, -self.num_prev_frames_memory:].transpose(1, 2).flatten(1, 2)  # n x T_prev*R x C
            d = torch.cat([f0, prev], dim=1)
            out.append(d.unsqueeze(2).repeat(1, 1, num_frames, 1))
        return out[1], out[0]

    def _cross_kv(self, mem, mem_key):
        """(k, v) per decoder layer: [HW_l, T, 256] column slices of one [HW_l, T, 256 n_l] projection per level and kind."""
        from ...layers import linear
        nl = self.num_feature_levels
        E = self.transformer_cross_attention_layers[0].multihead_attn.embed_dim
        mods = [layer.multihead_attn for layer in self.transformer_cross_attention_layers]
        key = tuple((m.in_proj_weight.data_ptr(), m.in_proj_weight._version, m.in_proj_bias._version) for m in mods) + (str(mem[0].device),)
        c = self.__dict__.get("_cross_kv_cache")
        if c is None or c[0] != key:
            with torch.no_grad():
                per_level = []
                for lvl in range(nl):
                    idx = list(range(lvl, self.num_layers, nl))
                    wk = torch.cat([mods[i].in_proj_weight[E:2 * E] for i in idx]).contiguous()
                    bk = torch.cat([mods[i].in_proj_bias[E:2 * E] for i in idx]).contiguous()
                    wv = torch.cat([mods[i].in_proj_weight[2 * E:] for i in idx]).contiguous()
                    bv = torch.cat([mods[i].in_proj_bias[2 * E:] for i in idx]).contiguous()
                    per_level.append((idx, wk, bk, wv, bv))
            c = (key, per_level)
            self.__dict__["_cross_kv_cache"] = c
        out = [None] * self.num_layers
        for lvl, (idx, wk, bk, wv, bv) in enumerate(c[1]):
            if not idx:
                continue
            k_all = linear(mem_key[lvl], wk, bk)
            v_all = linear(mem[lvl], wv, bv)
            for j, i in enumerate(idx):
                out[i] = (k_all[..., j * E:(j + 1) * E], v_all[..., j * E:(j + 1) * E])
        return out

    def _sa_mask_rows(self, mask, Qn, t_total, sl):
        """Rows of the [Q' T, Q' T] self-attention mask that belong to the frames `sl` (token order (q, t)): [Q' T_loc, Q' T]."""
        key = ("rows", id(mask), Qn, t_total, (sl.start, sl.stop) if isinstance(sl, slice) else tuple(sl))
        m = self._sa_mask_cache.get(key)
        if m is None or m[0] is not mask:
            rows = mask.view(Qn, t_total, Qn * t_total)[:, sl].reshape(-1, Qn * t_total).contiguous()
            m = (mask, rows)           # (the full mask is kept alive with its rows: id() keys must not be reused)
            self._sa_mask_cache[key] = m
        return m[1]

    def generate_self_attn_mask(self, bs, t, num_queries_lp, device, dataset_name, task):
        """:824-848: bool [QT, QT] (True = blocked), identical for every head (broadcast)."""
        tp = self.maskdec_self_attn_mask_type
        if tp in {"none", "all"}:
            return None
        blocked = tp == "sep-blocked" or task == "grounding"
        key = (t, num_queries_lp, str(device), tp, blocked)
        m = self._sa_mask_cache.get(key)
        if m is not None:
            return m
        nq = self.num_queries
        m = torch.ones((num_queries_lp * t, num_queries_lp * t), device=device, dtype=torch.bool)
        m[:nq * t, :nq * t] = False
        if blocked:
            n_p = num_queries_lp - nq
            for j in range(n_p):  # each prompt query only sees its own T copies
                s = (nq + j) * t
                m[s:s + t, s:s + t] = False
        elif tp == "sep":
            m[nq * t:, nq * t:] = False
        elif tp == "sep-l2p":
            m[nq * t:] = False
        else:
            raise ValueError(tp)
        if len(self._sa_mask_cache) > 16:
            self._sa_mask_cache.clear()
        self._sa_mask_cache[key] = m
        return m
