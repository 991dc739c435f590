"""Text / visual prompt encoders and the per-video prompt memory pool (inference).

Restates the inference branches of univs/modeling/prompt_encoder/prompt_encoder.py:
  TextPromptEncoder.get_expression_prompt (:28-55);
  VisualPromptEncoder.get_mask_prompt (:168-263), get_point_prompt (:82-165), get_box_prompt (:266-359),
  select_points_from_box_mask (:362-442), get_dense_features (:445-497);
  VisualPromptSampler.process_per_batch_inference (:782-842), process_per_video_inference (:845-960),
  process_per_video_inference_prev_frame (:963-1057), zero_pad_prompt (:1060-1071).

Contract kept bit-for-bit with the reference: the pool lives in the caller's
`targets[0]` dict and is mutated in place under the same keys (`prompt_feats`, `prompt_pe` [N_ent, R,
T_hist, C], `prompt_attn_masks` [T_hist, 1, N_ent, HW], `prompt_obj_ids`, `img_emb_per_video`,
`pos_emb_per_video`).  Random point selection uses `torch.randperm` on the default CPU generator in the
same call order as the reference, so a seeded run reproduces the reference's sampled tokens.
Known quirk kept on purpose: the "avoid NaN" block multiplies by `isblank` instead of `~isblank`
(:836-840), so blank prompt tokens are filled with zeros.
"""

import torch
import torch.nn.functional as F

from ..layers import point_sample, to_device_async
from ..switches import SWITCHES
from .position_encoding import PositionEmbeddingSine3D, PositionEmbeddingSine3DArbitraryT, _axis, _dim_t


# ---- box / mask helpers (univs/utils/comm.py:6-91) ---------------------------------------------------
def box_xyxy_to_cxcywh(x):
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, (x1 - x0), (y1 - y0)], dim=-1)


def convert_box_to_mask(outputs_box, h, w):
    """normalised xyxy boxes [..., 4] -> bool masks [..., h, w] (comm.py:6-39)."""
    box_shape = outputs_box.shape
    dev = outputs_box.device
    bx = outputs_box.flatten(0, -2)
    # (x * w, y * h) without materialising [w, h, w, h] on the host: a pageable H2D copy is a full stream sync
    b = torch.stack([(bx[..., 0] * w).floor(), (bx[..., 1] * h).floor(), (bx[..., 2] * w).ceil(), (bx[..., 3] * h).ceil()], dim=-1)
    gy, gx = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
    gy, gx = gy.reshape(1, h, w), gx.reshape(1, h, w)
    m = (gx > b[..., 0, None, None]) & (gx <= b[..., 2, None, None]) & \
        (gy > b[..., 1, None, None]) & (gy <= b[..., 3, None, None])
    return m.reshape((*box_shape[:-1], h, w))


def convert_mask_to_box(masks):
    """bool masks [..., H, W] -> xyxy pixel boxes [..., 4], zeros for empty masks (comm.py:41-84)."""
    if torch.numel(masks) == 0:
        return torch.zeros(*masks.shape[:-2], 4, device=masks.device)
    shape = masks.shape
    h, w = shape[-2:]
    masks = masks.flatten(0, -3) if len(shape) > 2 else masks.unsqueeze(0)
    in_h, _ = torch.max(masks, dim=-1)
    ch = in_h * torch.arange(h, device=in_h.device)[None, :]
    bottom, _ = torch.max(ch, dim=-1)
    top, _ = torch.min(ch + h * (~in_h), dim=-1)
    in_w, _ = torch.max(masks, dim=-2)
    cw = in_w * torch.arange(w, device=in_w.device)[None, :]
    right, _ = torch.max(cw, dim=-1)
    left, _ = torch.min(cw + w * (~in_w), dim=-1)
    empty = (right < left) | (bottom < top)
    out = torch.stack([left, top, right, bottom], dim=-1) * (~empty).unsqueeze(-1)
    return out.reshape(*shape[:-2], 4) if len(shape) > 2 else out[0]


def _ints_to_device(vals, device):
    """python ints / 0-dim tensors -> int64 [len] on `device` without a blocking copy (a pageable H2D copy waits for the
    whole stream; `int()` of a device tensor does too)."""
    if all(isinstance(v, torch.Tensor) and v.device == device for v in vals):
        return torch.stack([v.reshape(()) for v in vals]).to(torch.int64)
    if all(not (isinstance(v, torch.Tensor) and v.is_cuda) for v in vals):
        return to_device_async(torch.tensor([int(v) for v in vals], dtype=torch.int64), device)
    return torch.stack([torch.as_tensor(v).reshape(()).to(device, non_blocking=True) for v in vals]).to(torch.int64)


def _kth_true(mask, ranks):
    """mask [n, L] bool, ranks [n, R] int64 (0-based, < row count) -> flat index of the rank-th True of every
    row, in raster order -- `nonzero(mask[k])[rank]` for all rows at once, without the host round trip of
    `nonzero` (cumulative count + binary search)."""
    cs = torch.cumsum(mask, dim=1, dtype=torch.int32)
    return torch.searchsorted(cs, (ranks + 1).to(torch.int32), right=False).clamp(max=mask.shape[1] - 1)


def _kth_true_2d(mask, ranks, rowcnt=None):
    """Same for image-shaped masks [n, H, W] (full-resolution entity masks): first the image row that holds the
    rank-th True (per-row counts, a scan over H), then the column inside that row (a scan over W) -- two short scans
    instead of one over H*W elements per entity.  Returns flat indices y * W + x, [n, R]."""
    n, H, W = mask.shape
    rowcnt = mask.sum(2, dtype=torch.int32) if rowcnt is None else rowcnt
    rc = torch.cumsum(rowcnt, dim=1, dtype=torch.int32)                                   # [n, H] inclusive
    r1 = (ranks + 1).to(torch.int32)
    y = torch.searchsorted(rc, r1, right=False).clamp(max=H - 1)                             # [n, R]
    before = torch.where(y > 0, torch.gather(rc, 1, (y - 1).clamp(min=0)), torch.zeros_like(y, dtype=torch.int32))
    rows = torch.gather(mask, 1, y[:, :, None].expand(-1, -1, W))                           # [n, R, W]
    cs = torch.cumsum(rows, dim=2, dtype=torch.int32)
    x = torch.searchsorted(cs, (r1 - before)[:, :, None].contiguous(), right=False).squeeze(-1).clamp(max=W - 1)
    return y * W + x


class TextPromptEncoder:
    """Referring expressions -> the text prompts of the hot path (prompt_encoder.py:16-55):
    `exp_word_feats [E, 77, T, 640]` = per-token features of the bare expression ('{}.' template),
    `exp_sentence_feats [E, T, 640]` = end-of-text feature averaged over the 81 prompt templates, both replicated over the
    clip's frames, and `len_word_expressions` = words + 5.  The reference encodes one expression (81 x 77 tokens) per
    call; here all E x 81 texts go through the text transformer as one batch."""

    def __init__(self, lang_encoder, num_frames, device=None):
        self.lang_encoder = lang_encoder
        if lang_encoder is not None:
            self.lang_encoder = lang_encoder.to(device or torch.device("cuda" if torch.cuda.is_available() else "cpu"))
        self.num_frames = num_frames

    @torch.no_grad()
    def get_expression_prompt(self, expressions, device, tokens=None, max_batch=4096):
        """`tokens` (optional): pre-tokenized `[E, n_templates, 77]` ids, e.g. cached per video."""
        assert self.lang_encoder is not None, "No language encoder is assigned!!"
        from .language import pre_tokenize_expression
        len_word_expressions = [len(exp.split(" ")) + 5 for exp in expressions]
        if tokens is None:
            tokens = pre_tokenize_expression(expressions)
        tokens = tokens.to(device)
        E, P, L = tokens.shape
        flat = tokens.reshape(E * P, L)
        words, eots = [], []
        for lo in range(0, E * P, max_batch):            # bounded activation memory for long expression lists
            w, e = self.lang_encoder.encode_text(flat[lo:lo + max_batch], only_eot=False)
            first = [i - lo for i in range(lo, min(lo + max_batch, E * P)) if i % P == 0]
            if first:                                    # only the bare-expression rows are consumed per token
                words.append(w[torch.as_tensor(first, device=w.device)])
            eots.append(e)
        exp_word_feats = torch.cat(words)                                   # [E, 77, C]
        exp_sentence_feats = torch.cat(eots).view(E, P, -1).mean(1)         # [E, C]
        exp_word_feats = exp_word_feats[:, :, None].repeat(1, 1, self.num_frames, 1)
        exp_sentence_feats = exp_sentence_feats[:, None].repeat(1, self.num_frames, 1)
        return exp_word_feats, exp_sentence_feats, len_word_expressions


class VisualPromptEncoder:
    def __init__(self, pretrain_img_size=1024, hidden_dim=256, num_frames=1, num_dense_points=32,
                 position_embedding_sin3d_type="FixedT"):
        N_steps = hidden_dim // 2
        if position_embedding_sin3d_type == "FixedT":
            self.pe_layer = PositionEmbeddingSine3D(N_steps, normalize=True)
        else:
            self.pe_layer = PositionEmbeddingSine3DArbitraryT(N_steps, normalize=True)
        self.num_frames = num_frames
        self.pretrain_img_size = pretrain_img_size
        self.num_dense_points = num_dense_points
        self.position_embedding_sin3d_type = position_embedding_sin3d_type
        self.key_fid = int((num_frames - 1) / 2)
        self.img_feats_scale = 8  # prompts are read from the 1/8-resolution level
        # frame-sharded clips (univs_amd/distributed.py): a sum over the ranks.  Only the rank that owns a key frame
        # holds its features, the others pass zeros; token features are linear in them, so the sum hands every rank
        # the owner's values (x + 0 is exact).  Everything else (candidate pixels, random ranks, position tokens,
        # attention masks) depends on the replicated annotations only and is evaluated identically on every rank.
        self.feature_reduce = None
        # "reference" (default): the reference's own `torch.randperm` draws on the CPU generator, in its call order --
        # bit-identical sampling, at the price of one host round trip per key frame (the draw sizes are pixel counts that
        # live on the device).  "device" (UNIVS_SAMPLER=device): the same distributions drawn on the device (uniform
        # rank per point; uniform R-subset in random order via top-R of random keys), no host round trip; the random
        # stream differs from the reference's, everything that is not random (cyclic fill of small masks, fallbacks of
        # empty ones) is identical.
        self.sampler_rng = SWITCHES.sampler          # "auto" | "reference" | "device" (switches.py; settable per encoder)
        if self.sampler_rng not in ("auto", "reference", "device"):
            raise ValueError(f"sampler mode {self.sampler_rng!r} (expected 'auto', 'reference' or 'device')")
        self._dev_gen = {}
        self._replay = None
        # a list: every get_mask_prompt call (or key frame of get_mask_prompts) appends the pixels it sampled, in the format
        # of `set_replay` -- (point_idx [n] int32, feat_idx [n, R] int32 with -1 rows for empty masks), on the host
        self.draw_log = None

    def set_replay(self, draws):
        """Replay recorded draws instead of drawing: `draws` = a sequence of (point_idx [n], feat_idx [n, R]) per
        `get_mask_prompt` call, in call order -- the sampled point as a flat index y * w + x of the full-resolution mask and
        the R dense-token pixels as flat feature-map indices (-1 = empty mask); None switches replay off.  The SIZES of the
        reference's `randperm` draws are pixel counts of thresholded masks (prompt_encoder.py:420,424,481), so a run whose
        mask differs from a recorded run in one near-threshold pixel cannot reproduce that run's tokens from the seed;
        replaying the sampled pixels can (deterministic re-runs, and parity runs against draws recorded inside the
        reference: tests/golden/g20, oracle/gen_golden.py:_capture_sampler_draws).  No host round trip in this mode."""
        import collections
        self._replay = None if draws is None else collections.deque(draws)

    def replay_pending(self):
        return 0 if self._replay is None else len(self._replay)

    def _rng(self, where):
        """The sampler mode in effect for tensors on `where` (a tensor or a device): "auto" = "device" on the GPU (draws from the device
        generator: no host round trip, no host-side randperm over an entity's candidate pixels -- 4 ms each for a large mask at 720p,
        82 -> 8 ms per clip in profiles/r05_video_loop_stages_v1.txt) and "reference" on the CPU (the reference's host `randperm`
        calls in the reference's order: its random stream, draw for draw)."""
        if self.sampler_rng != "auto":
            return self.sampler_rng
        dev = where.device if isinstance(where, torch.Tensor) else torch.device(where)
        return "device" if dev.type == "cuda" else "reference"

    def _generator(self, device):
        g = self._dev_gen.get(str(device))
        if g is None:
            g = torch.Generator(device=device)
            seed = getattr(self, "_video_seed", None)
            g.manual_seed((torch.initial_seed() if seed is None else seed) % (2 ** 63))
            self._dev_gen[str(device)] = g
        return g

    def begin_video(self, device, shard=None):
        """Called by the clip loops at the start of every video (inference/video_entity.py, video_vos.py).
        * "device" draws (the default for GPU tensors): the device generators are reseeded from ONE value drawn from the default
          (CPU) generator, so `torch.manual_seed(s)` in front of a video makes its prompts reproducible on the GPU whatever ran
          before it (without this the generator was seeded once per process and a video's draws depended on how many videos came
          first: ADVICE r05).  The "reference" mode draws nothing here: its stream stays the reference's, draw for draw.
        * `shard` (a frame-sharded loop, univs_amd.distributed.FrameShard: every rank runs the sampler on the replicated state and
          must draw the SAME numbers): the state of rank 0's default generator is broadcast and installed on every rank of the
          group first -- rank 0's stream is untouched (the sharded video equals the single-process video with rank 0's seed), and
          rank-dependent seeding (detectron2's seed + rank) can no longer let the ranks sample different pixels and mix them in
          the next collective."""
        sharded = shard is not None and shard.world > 1
        if sharded:
            import torch.distributed as dist
            state = torch.get_rng_state().to(device)
            src = dist.get_global_rank(shard.group, 0) if shard.group is not None else 0
            dist.broadcast(state, src=src, group=shard.group)
            torch.set_rng_state(state.cpu())
        if self._rng(device) == "reference":
            return None
        seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)            # the default (CPU) generator: the same on every rank
        self._video_seed = int(seed.item())
        for g in self._dev_gen.values():
            g.manual_seed(self._video_seed % (2 ** 63))
        return self._video_seed

    def _point_pe(self, h_img, w_img, point_coords, key_fid, key_fid_original):
        size = (self.num_frames, h_img * self.img_feats_scale, w_img * self.img_feats_scale)
        if self.position_embedding_sin3d_type == "FixedT":
            pe = self.pe_layer.forward_points_with_size(size, point_coords)
            return pe[key_fid].unsqueeze(1).repeat(1, self.num_frames, 1)
        return self.pe_layer.forward_points_with_size(size, point_coords, key_fid_original).transpose(0, 1)

    @torch.no_grad()
    def get_point_prompt(self, img_features, img_pos, point_coords=None, boxes=None, masks=None, key_fid=None,
                         key_fid_original=None, is_train=False, enable_dense_prompt=True):
        key_fid = self.key_fid if key_fid is None else key_fid
        key_fid_original = key_fid if key_fid_original is None else key_fid_original
        h_img, w_img = img_features.shape[-2:]
        if point_coords is None:
            assert boxes is not None or masks is not None
            point_coords = self.select_points_from_box_mask(h_img, w_img, masks=masks, boxes=boxes)
        device = img_features.device
        point_coords = point_coords.to(device)
        valid = ((point_coords >= 0) & ((1 - point_coords) >= 0)).sum(-1) == 2
        point_coords = point_coords * valid.float().view(-1, 1)
        n = point_coords.shape[0]
        query_pe = self._point_pe(h_img, w_img, point_coords, key_fid, key_fid_original)
        point_feats = point_sample(img_features.unsqueeze(0), point_coords.unsqueeze(0), align_corners=False)
        query_feats = point_feats.permute(2, 0, 1).repeat(1, self.num_frames, 1)
        masks_ = torch.ones((self.num_frames, 1, n, h_img * w_img), dtype=torch.bool, device=device)
        wh = point_coords * torch.as_tensor([w_img, h_img], device=device).view(1, -1)
        fl, ce = wh.floor().long(), wh.ceil().long()
        rng = torch.arange(n, device=device)
        for yy, xx in ((fl[:, 1], fl[:, 0]), (ce[:, 1], ce[:, 0]), (fl[:, 1], ce[:, 0]), (ce[:, 1], fl[:, 0])):
            idx = (yy * w_img + xx).clamp(min=0, max=w_img * h_img - 1)
            masks_[key_fid, :, rng, idx] = False
        fd, pd = query_feats[:, None], query_pe[:, None]
        if enable_dense_prompt:
            fd = fd.repeat(1, self.num_dense_points, 1, 1)
            pd = pd.repeat(1, self.num_dense_points, 1, 1)
        if self.feature_reduce is not None:
            # frame-sharded mode: only the rank that owns the key frame holds its features (zeros elsewhere); the token
            # features are linear in them, so their sum over the ranks is the single-process value on every rank
            fd = self.feature_reduce(fd[:, :, 0].contiguous())[:, :, None].repeat(1, 1, fd.shape[2], 1)
        if (~valid).any():
            pd = pd * valid.view(-1, 1, 1, 1)
            fd = fd * valid.view(-1, 1, 1, 1)
            masks_[:, :, ~valid] = False
        return point_coords, pd, fd, masks_

    @torch.no_grad()
    def annotation_prefix(self, masks, boxes, h_img, w_img, mask_thresh=0.5):
        """The part of `get_mask_prompt` that depends on the annotations only -- not on the image features -- for F key
        frames at once: masks [F, n, h, w], boxes [F, n, 4] or None -> tensors with a leading frame axis: `valid` / `visible` [F, n] (some pixel above the threshold / above 0),
        `feat_masks` / `feat_masks_binary` [F, n, h_img, w_img], the point candidates `sel` [F, n, h, w] with their row
        counts `rowcnt` [F, n, h], and `counts` [F, 2 n] int32 = the sizes of the reference's randperm draws of one call
        (points of the n entities, then dense tokens of the n entities).  Every quantity is per entity except the
        threshold of `feat_masks_binary`, which is per frame (prompt_encoder.py:198-200 takes the max over the call)."""
        Fk, n, h, w = masks.shape
        s = self.img_feats_scale
        assert (h_img * s == h) and (w_img * s == w), \
            f"Input images must have same size with masks: {(h, w), (h_img * s, w_img * s)}"
        if SWITCHES.fused_sampler and masks.is_cuda and boxes is not None and Fk * n > 0 and mask_thresh == 0.5:
            # three launches instead of ~80 (csrc/prompt_sampler.hip), the same values bit for bit
            from .. import ops
            return ops.prompt_prefix(masks.float(), boxes.float().to(masks.device), s, mask_thresh)
        flat = masks.reshape(Fk * n, h, w)
        mx = flat.amax(2).amax(1)                             # two short reductions
        valid = mx > mask_thresh                              # some pixel above the threshold
        feat_masks = F.interpolate(flat.float().unsqueeze(1), (h_img, w_img), mode="nearest").squeeze(1)
        thr = feat_masks.reshape(Fk, -1).amax(1).clamp(max=mask_thresh)
        feat_masks_binary = feat_masks >= thr.view(Fk, 1, 1, 1).expand(Fk, n, 1, 1).reshape(-1, 1, 1)
        sel, rowcnt = self._select_candidates(flat, None if boxes is None else boxes.reshape(Fk * n, 4))
        counts = torch.cat([rowcnt.sum(1, dtype=torch.int32).view(Fk, n),
                            feat_masks_binary.flatten(1).sum(1, dtype=torch.int32).view(Fk, n)], dim=1)
        return {"valid": valid.view(Fk, n), "visible": (mx > 0).view(Fk, n), "feat_masks": feat_masks.view(Fk, n, h_img, w_img),
                "feat_masks_binary": feat_masks_binary.view(Fk, n, h_img, w_img), "sel": sel.view(Fk, n, h, w),
                "rowcnt": rowcnt.view(Fk, n, h), "counts": counts}

    @torch.no_grad()
    def get_mask_prompt(self, img_features, img_pos, masks, boxes=None, mask_thresh=0.5, key_fid=None,
                        key_fid_original=None, is_train=False, enable_dense_prompt=True, _pre=None, _counts=None):
        """`_pre`: this call's slice of `annotation_prefix` (computed ahead for all key frames of a clip by
        VisualPromptSampler.prefetch); `_counts`: its `counts` row already on the host (list of 2 n ints)."""
        key_fid = self.key_fid if key_fid is None else key_fid
        key_fid_original = key_fid if key_fid_original is None else key_fid_original
        h_img, w_img = img_features.shape[-2:]
        device = img_features.device
        assert masks.dim() == 3, f"Mask shape shoule be num_instsxHxW, but get {masks.shape}"
        n, h, w = masks.shape
        s = self.img_feats_scale
        if _pre is None:
            _pre = self.annotation_prefix(masks[None], None if boxes is None else boxes[None], h_img, w_img, mask_thresh)
            _pre = {k: v[0] for k, v in _pre.items()}
        valid, feat_masks, feat_masks_binary = _pre["valid"], _pre["feat_masks"], _pre["feat_masks_binary"]
        replay_feat_idx = None
        if self._replay is not None:
            assert self._replay, "sampler replay: more get_mask_prompt calls than recorded draws"
            point_idx, replay_feat_idx = self._replay.popleft()
            assert point_idx.shape[0] == n and replay_feat_idx.shape[0] == n, "sampler replay: entity count differs from the recording"
            point_idx = to_device_async(point_idx.to(torch.int64), device)
            replay_feat_idx = to_device_async(replay_feat_idx.to(torch.int64), device)
            counts = [None] * (2 * n)
            point_coords = torch.stack([((point_idx % w).float() + 0.5) / w, ((point_idx // w).float() + 0.5) / h], dim=-1)
        elif self._rng(masks) == "device":
            counts = [None] * (2 * n)                        # sizes stay on the device
            point_coords = self.select_points_from_box_mask(h_img, w_img, masks=masks, boxes=boxes,
                                                            _prepared=(_pre["sel"], _pre["rowcnt"], None))
        else:
            # the pixel counts that size the reference's randperm calls (point selection, then dense tokens -- generated
            # on the host in exactly that order): ONE host round trip per call unless the caller fetched them ahead
            counts = _pre["counts"].tolist() if _counts is None else _counts
            point_coords = self.select_points_from_box_mask(h_img, w_img, masks=masks, boxes=boxes,
                                                            _prepared=(_pre["sel"], _pre["rowcnt"], counts[:n]))
        query_pe = self._point_pe(h_img, w_img, point_coords, key_fid, key_fid_original)
        fw = feat_masks * feat_masks_binary
        pf = torch.einsum("qn,nc->qc", fw.flatten(-2).float(), img_features.flatten(-2).t())
        pf = pf / fw.sum((-2, -1)).clamp(min=mask_thresh)[:, None]
        query_feats = pf[:, None].repeat(1, self.num_frames, 1)
        if boxes is None:
            # the reference builds this normaliser on the host and divides a device tensor by it
            # (prompt_encoder.py:243-244: device mismatch on a GPU); same values, on the masks' device
            normlizer = torch.tensor([w_img * s, h_img * s, w_img * s, h_img * s], dtype=torch.float32,
                                     device=masks.device).reshape(1, -1)
            boxes = convert_mask_to_box(masks > mask_thresh) / normlizer
        attn = torch.zeros((self.num_frames, 1, n, h_img * w_img), dtype=torch.bool, device=device)
        attn[key_fid, 0] = torch.logical_not(convert_box_to_mask(boxes, h_img, w_img).flatten(-2))
        fd, pd = query_feats[:, None], query_pe[:, None]
        if enable_dense_prompt:
            fd, pd = self.get_dense_features(img_features, img_pos, feat_masks_binary, query_pe, query_feats,
                                             prompt_type="masks", is_train=is_train,
                                             _counts=None if self._rng(masks) == "device" else counts[n:],
                                             _replay_idx=replay_feat_idx)
        if self.feature_reduce is not None:
            fd = self.feature_reduce(fd[:, :, 0].contiguous())[:, :, None].repeat(1, 1, fd.shape[2], 1)
        if self.draw_log is not None and enable_dense_prompt:
            px = torch.round(point_coords[:, 0] * w - 0.5).long()
            py = torch.round(point_coords[:, 1] * h - 0.5).long()
            self._log_draws((py * w + px)[None], self._last_dense[0][None], self._last_dense[1][None])
        # invalid (empty) entities: zero tokens, nothing masked (unconditional: no host round trip for `.any()`)
        pd = pd * valid.view(-1, 1, 1, 1).float()
        fd = fd * valid.view(-1, 1, 1, 1).float()
        attn = attn & valid.view(1, 1, -1, 1)
        return point_coords, pd, fd, attn

    def _log_draws(self, point_idx, dense_idx, empty):
        """[F, n], [F, n, R], [F, n] -> one (point_idx, feat_idx) record per key frame"""
        pi = point_idx.to(torch.int32).cpu()
        di = torch.where(empty[..., None], torch.full_like(dense_idx, -1), dense_idx).to(torch.int32).cpu()
        for f in range(pi.shape[0]):
            self.draw_log.append((pi[f], di[f]))

    @torch.no_grad()
    def get_mask_prompts(self, img_features, img_pos, masks, boxes, key_fids, key_fids_original, pre, counts=None,
                         mask_thresh=0.5):
        """`get_mask_prompt` for F key frames in ONE pass over the device (a prompted clip re-encodes T - clip_stride key
        frames: ~110 launches each when called one by one).  img_features / img_pos [F, C, h_img, w_img] = the key frames'
        maps, masks [F, n, h, w], boxes [F, n, 4], key_fids / key_fids_original = F clip-relative / absolute frame
        indices, `pre` = annotation_prefix(masks, boxes), `counts` = its `counts` rows on the host (F lists of 2 n ints;
        "reference" sampler only).  Returns (point_coords [F, n, 2], pd, fd [F, n, R, T, C], attn [F, T, 1, n, h_img*w_img]):
        slice f is what the f-th separate call returns.  The host draws are generated frame by frame -- points of frame
        0, dense tokens of frame 0, points of frame 1, ... -- i.e. in the order of F separate calls (and of the
        reference's loop, prompt_encoder.py:463-474 / :236-251), so a seeded run samples the same pixels."""
        assert self.feature_reduce is None, "frame-sharded clips go through get_mask_prompt (one reduction per key frame)"
        Fk, n, h, w = masks.shape
        C = img_features.shape[1]
        h_img, w_img = img_features.shape[-2:]
        HW, T, R = h_img * w_img, self.num_frames, self.num_dense_points
        device = img_features.device
        N = Fk * n
        valid, feat_masks, fmb = pre["valid"].reshape(N), pre["feat_masks"], pre["feat_masks_binary"]
        m = fmb.reshape(N, HW)
        # ---- the draws: a rank among the candidate pixels per entity, R ranks among the mask's feature pixels per entity
        dense_idx = point_coords = drawn = None
        fused = SWITCHES.fused_sampler and img_features.is_cuda and all(pre[k].is_contiguous() for k in ("sel", "rowcnt", "feat_masks_binary", "counts"))
        if fused:
            from .. import ops
        if self._replay is not None:
            assert len(self._replay) >= Fk, "sampler replay: more get_mask_prompt calls than recorded draws"
            rec = [self._replay.popleft() for _ in range(Fk)]
            assert all(r[0].shape[0] == n and tuple(r[1].shape) == (n, R) for r in rec), \
                "sampler replay: entity count differs from the recording"
            point_idx = to_device_async(torch.cat([r[0].to(torch.int64) for r in rec]), device)
            dense_idx = to_device_async(torch.cat([r[1].to(torch.int64) for r in rec]), device)
            empty = (dense_idx[:, :1] < 0).view(-1, 1, 1)
            dense_idx = dense_idx.clamp(min=0)
        elif self._rng(device) == "device":
            # two draws per call, in this order, whichever formulation turns them into pixels
            u = torch.rand((N, 1), device=device, generator=self._generator(device))
            keys = torch.rand(m.shape, device=device, generator=self._generator(device))
            drawn = ops.prompt_draw(pre, R, u=u, keys=keys) if fused else None
        if dense_idx is not None:
            pass                                                                # replayed pixels
        elif drawn is not None:
            point_idx, point_coords, dense_idx, empty = drawn
            empty = empty.view(-1, 1, 1)
        elif self._rng(device) == "device":
            rowcnt = pre["rowcnt"].reshape(N, h)
            cnt = rowcnt.sum(1, dtype=torch.int64).clamp(min=1)
            ranks = (u * cnt[:, None]).long().clamp(max=cnt[:, None] - 1)
            point_idx = _kth_true_2d(pre["sel"].reshape(N, h, w), ranks, rowcnt)[:, 0]
            dcnt = m.sum(1)
            cyc = torch.arange(R, device=device)[None] % dcnt.clamp(min=1)[:, None]
            idx_small = _kth_true(m, cyc)
            keys = keys.masked_fill(~m.bool(), -1.0)
            idx_big = keys.topk(min(R, HW), dim=1).indices
            if idx_big.shape[1] < R:
                idx_big = torch.cat([idx_big, idx_big[:, :1].expand(-1, R - idx_big.shape[1])], dim=1)
            dense_idx = torch.where((dcnt >= R)[:, None], idx_big, idx_small)
            empty = (dcnt == 0).view(-1, 1, 1)
        else:
            assert counts is not None and len(counts) == Fk
            rows_p, rows_d = [], []
            for f in range(Fk):
                rows_p += [torch.randperm(int(c))[:1] for c in counts[f][:n]]
                for c in counts[f][n:]:
                    c = int(c)
                    if c == 0:
                        rows_d.append(torch.zeros(R + 1, dtype=torch.int64))
                        rows_d[-1][R] = 1                                       # last column: the entity is empty
                    elif c < R:
                        rows_d.append(torch.cat([torch.arange(c).repeat(int(R / c) + 1)[:R], torch.zeros(1, dtype=torch.int64)]))
                    else:
                        rows_d.append(torch.cat([torch.randperm(c)[:R], torch.zeros(1, dtype=torch.int64)]))
            tab = to_device_async(torch.cat([torch.stack(rows_d), torch.stack(rows_p)], dim=1), device)   # one transfer
            drawn = ops.prompt_draw(pre, R, tab=tab) if fused else None
            if drawn is not None:
                point_idx, point_coords, dense_idx, empty = drawn
                empty = empty.view(-1, 1, 1)
            else:
                point_idx = _kth_true_2d(pre["sel"].reshape(N, h, w), tab[:, R + 1:], pre["rowcnt"].reshape(N, h))[:, 0]
                dense_idx = _kth_true(m, tab[:, :R])
                empty = (tab[:, R] != 0).view(-1, 1, 1)
        if self.draw_log is not None:
            self._log_draws(point_idx.view(Fk, n), dense_idx.view(Fk, n, R), empty.view(Fk, n))
        if point_coords is None:
            point_coords = torch.stack([((point_idx % w).float() + 0.5) / w, ((point_idx // w).float() + 0.5) / h], dim=-1)
        # ---- position token of the sampled point at its key frame, replicated over the clip's frames
        kf = torch.arange(Fk, device=device) if list(key_fids) == list(range(Fk)) else _ints_to_device(key_fids, device)
        size = (T, h_img * self.img_feats_scale, w_img * self.img_feats_scale)
        query_pe = None
        if fused:
            # one launch for the F n position tokens (csrc/prompt_sampler.hip: ps_point_pe; the ATen formulation below makes them for
            # every (frame, point) pair in ~25 launches and keeps the diagonal), the same bits
            pl = self.pe_layer
            if self.position_embedding_sin3d_type == "FixedT":
                zf = _axis(T, pl.scale, device)[kf]
            else:
                zf = _ints_to_device(key_fids_original, device) / pl.num_max_frames * pl.scale
            query_pe = ops.prompt_point_pe(point_coords.reshape(N, 2), zf.float(), _dim_t(pl.num_pos_feats, pl.temperature, device),
                                           _dim_t(2 * pl.num_pos_feats, pl.temperature, device), pl.scale, n)
        if query_pe is not None:
            pass
        elif self.position_embedding_sin3d_type == "FixedT":
            pe = self.pe_layer.forward_points_with_size(size, point_coords)                        # [T, N, C]
            query_pe = pe.view(T, Fk, n, -1)[kf, torch.arange(Fk, device=device)]                 # [F, n, C]
        else:
            kfo = _ints_to_device(key_fids_original, device)
            z = kfo / self.pe_layer.num_max_frames * self.pe_layer.scale
            ar = torch.arange(Fk, device=device)
            query_pe = self.pe_layer._points(z, point_coords).view(Fk, Fk, n, -1)[ar, ar]
        # ---- mask-pooled feature token
        fw = feat_masks * fmb                                                                       # [F, n, h_img, w_img]
        feats = img_features.flatten(-2).transpose(1, 2)                                            # [F, HW, C]
        pf = torch.bmm(fw.flatten(-2).float(), feats) / fw.sum((-2, -1)).clamp(min=mask_thresh)[..., None]
        if fused:
            # the dense tokens and the cross-attention masks: two launches (csrc/prompt_sampler.hip)
            out = ops.prompt_tokens(img_features, img_pos, pf.reshape(N, C), query_pe.reshape(N, -1), dense_idx, empty.view(-1), valid,
                                    boxes.float().to(device), kf, T)
            if out is not None:
                return point_coords.view(Fk, n, 2), out[1], out[0], out[2]
        query_pe = query_pe.reshape(N, 1, -1).repeat(1, T, 1)                                      # [N, T, C]
        query_feats = pf.reshape(N, 1, C).repeat(1, T, 1)
        # ---- cross-attention mask: everything outside the box, at the key frame only
        attn = torch.zeros((Fk, T, 1, n, HW), dtype=torch.bool, device=device)
        attn[torch.arange(Fk, device=device), kf, 0] = torch.logical_not(convert_box_to_mask(boxes, h_img, w_img).flatten(-2))
        # ---- dense tokens: features / position embeddings at the R sampled pixels of the key frame's map
        off = (torch.arange(Fk, device=device) * HW).repeat_interleave(n)[:, None]
        gidx = dense_idx + off
        fd = torch.where(empty, query_feats[:, 0][:, None].expand(-1, R, -1), feats.reshape(Fk * HW, C)[gidx])
        pd = torch.where(empty, query_pe[:, 0][:, None].expand(-1, R, -1),
                         img_pos.flatten(-2).transpose(1, 2).reshape(Fk * HW, -1)[gidx])
        vf = valid.view(-1, 1, 1, 1).float()
        fd = (fd[:, :, None].repeat(1, 1, T, 1) * vf).view(Fk, n, R, T, -1)
        pd = (pd[:, :, None].repeat(1, 1, T, 1) * vf).view(Fk, n, R, T, -1)
        attn = attn & valid.view(Fk, 1, 1, n, 1)
        return point_coords.view(Fk, n, 2), pd, fd, attn

    @torch.no_grad()
    def get_box_prompt(self, img_features, img_pos, boxes, key_fid=None, key_fid_original=None, is_train=False,
                       enable_dense_prompt=True):
        key_fid = self.key_fid if key_fid is None else key_fid
        key_fid_original = key_fid if key_fid_original is None else key_fid_original
        h_img, w_img = img_features.shape[-2:]
        device = img_features.device
        assert boxes.dim() == 2
        valid = (box_xyxy_to_cxcywh(boxes)[..., 2:] > 0).all(-1)
        point_coords = self.select_points_from_box_mask(h_img, w_img, boxes=boxes)
        query_pe = self._point_pe(h_img, w_img, point_coords, key_fid, key_fid_original)
        box_masks = convert_box_to_mask(boxes, h_img, w_img)
        qf = (box_masks[:, None] * img_features[None]).flatten(-2).sum(-1) / \
            box_masks[:, None].flatten(-2).sum(-1).clamp(min=1)
        blank = box_masks.flatten(-2).sum(-1) == 0
        if blank.any():
            qf[blank] = point_sample(img_features.unsqueeze(0), point_coords[blank].unsqueeze(0),
                                     align_corners=False).squeeze(0).t()
        query_feats = qf[:, None].repeat(1, self.num_frames, 1)
        attn = torch.zeros((self.num_frames, 1, box_masks.shape[0], h_img * w_img), dtype=torch.bool, device=device)
        attn[key_fid, 0] = torch.logical_not(box_masks.flatten(-2))
        fd, pd = query_feats[:, None], query_pe[:, None]
        if enable_dense_prompt:
            fd, pd = self.get_dense_features(img_features, img_pos, box_masks, query_pe, query_feats, is_train=is_train)
        if self.feature_reduce is not None:   # frame-sharded mode (see get_point_prompt)
            fd = self.feature_reduce(fd[:, :, 0].contiguous())[:, :, None].repeat(1, 1, fd.shape[2], 1)
        if (~valid).any():
            pd = pd * valid.view(-1, 1, 1, 1)
            fd = fd * valid.view(-1, 1, 1, 1)
            attn[:, :, ~valid] = False
        return point_coords, pd, fd, attn

    @torch.no_grad()
    def _select_candidates(self, masks, boxes, mask_thresh=0.75):
        """Device part of the mask branch of `select_points_from_box_mask`: the candidate pixels of every entity
        ([n, h, w] bool) and their per-row counts ([n, h] int32)."""
        n, h, w = masks.shape
        device = masks.device
        masks = masks.float()
        if boxes is None:
            boxes = convert_mask_to_box(masks > mask_thresh) / torch.as_tensor([w, h, w, h], device=device).view(1, -1)
        bc = box_xyxy_to_cxcywh(boxes).to(device)
        mx = masks.amax(2).amax(1)                            # two-stage: rows of H*W elements reduce slowly
        masks_binary = masks >= mx.clamp(max=mask_thresh).view(-1, 1, 1)
        # pixel centres within the central half of the box (|c - centre| < w/4, h/4), separably
        xs = (torch.arange(w, device=device, dtype=torch.float32) + 0.5) / w
        ys = (torch.arange(h, device=device, dtype=torch.float32) + 0.5) / h
        in_x = torch.abs(xs[None] - bc[:, None, 0]) < 0.25 * bc[:, None, 2]
        in_y = torch.abs(ys[None] - bc[:, None, 1]) < 0.25 * bc[:, None, 3]
        in_ctr = in_y[:, :, None] & in_x[:, None, :] & masks_binary
        # entities without a central pixel fall back to their most confident pixels
        hi = masks >= mx.clamp(max=0.95).view(-1, 1, 1)
        sel = torch.where(in_ctr.any(2).any(1).view(-1, 1, 1), in_ctr, hi)
        return sel, sel.sum(2, dtype=torch.int32)

    @torch.no_grad()
    def select_points_from_box_mask(self, h_img, w_img, boxes=None, masks=None, is_train=False, mask_thresh=0.75,
                                    num_points=1, _prepared=None):
        assert (boxes is not None) or (masks is not None)
        assert not is_train
        if masks is not None:
            device = masks.device
            n, h, w = masks.shape
            masks = masks.float()
            s = self.img_feats_scale
            assert (h_img * s == h) and (w_img * s == w), \
                f"Input images must have same size with masks: {(h, w), (h_img * s, w_img * s)}"
            if _prepared is None:
                sel, rowcnt = self._select_candidates(masks, boxes, mask_thresh)
                counts = None if self._rng(masks) == "device" else rowcnt.sum(1).tolist()   # the one host round trip
            else:
                sel, rowcnt, counts = _prepared                   # get_mask_prompt shares one round trip
            if self._rng(masks) == "device":
                # a uniform candidate per point: rank = floor(u * count), count >= 1 by construction of `sel`
                cnt = rowcnt.sum(1, dtype=torch.int64).clamp(min=1)
                u = torch.rand((n, num_points), device=device, generator=self._generator(device))
                ranks = (u * cnt[:, None]).long().clamp(max=cnt[:, None] - 1)
            else:
                # same generator calls, in the same order, as the reference's per-entity loop (prompt_encoder.py:463-474)
                ranks = to_device_async(torch.stack([torch.randperm(int(c)).repeat(num_points)[:num_points] for c in counts]), device)
            idx = _kth_true_2d(sel, ranks, rowcnt)                 # [n, num_points] flat pixel indices
            point_coords = torch.stack([((idx % w).float() + 0.5) / w, ((idx // w).float() + 0.5) / h], dim=-1)
        else:
            device = boxes.device
            bc = box_xyxy_to_cxcywh(boxes)
            cxcy = bc[:, :2][:, None].repeat(1, num_points, 1)
            wh = bc[:, 2:][:, None].repeat(1, num_points, 1)
            offsets = torch.rand(wh.shape).to(device) * 2 - 1
            point_coords = cxcy + offsets * 0.25 * wh
        return point_coords[:, 0] if num_points == 1 else point_coords

    @torch.no_grad()
    def get_dense_features(self, img_features, img_pos, masks_binary, query_pe, query_feats, prompt_type="masks",
                           is_train=True, _counts=None, _replay_idx=None):
        assert img_features.shape[-2:] == masks_binary.shape[-2:]
        feats = img_features.flatten(-2).t()
        pos = img_pos.flatten(-2).t()
        R = self.num_dense_points
        m = masks_binary.flatten(1)
        if _replay_idx is not None:
            assert tuple(_replay_idx.shape) == (m.shape[0], R), "sampler replay: dense-token table has the wrong shape"
            empty = (_replay_idx[:, :1] < 0).view(-1, 1, 1)
            idx = _replay_idx.clamp(min=0)
            self._last_dense = (idx, empty.view(-1))
            fd = torch.where(empty, query_feats[:, 0][:, None].expand(-1, R, -1), feats[idx])
            pd = torch.where(empty, query_pe[:, 0][:, None].expand(-1, R, -1), pos[idx])
            return (fd[:, :, None].repeat(1, 1, self.num_frames, 1), pd[:, :, None].repeat(1, 1, self.num_frames, 1))
        if self._rng(m) == "device":
            assert not (prompt_type == "masks" and is_train), "training branch is out of scope"
            cnt = m.sum(1)                                                        # [n] on the device
            # masks with fewer than R pixels: all of them, cyclically, in pixel order (the reference's rule, :240-246)
            cyc = torch.arange(R, device=m.device)[None] % cnt.clamp(min=1)[:, None]
            idx_small = _kth_true(m, cyc)
            # larger masks: a uniform R-subset in random order = the R largest of i.i.d. keys on the mask's pixels
            keys = torch.rand(m.shape, device=m.device, generator=self._generator(m.device)).masked_fill(~m.bool(), -1.0)
            idx_big = keys.topk(min(R, m.shape[1]), dim=1).indices
            if idx_big.shape[1] < R:
                idx_big = torch.cat([idx_big, idx_big[:, :1].expand(-1, R - idx_big.shape[1])], dim=1)
            idx = torch.where((cnt >= R)[:, None], idx_big, idx_small)
            empty = (cnt == 0).view(-1, 1, 1)
            self._last_dense = (idx, empty.view(-1))
            fd = torch.where(empty, query_feats[:, 0][:, None].expand(-1, R, -1), feats[idx])
            pd = torch.where(empty, query_pe[:, 0][:, None].expand(-1, R, -1), pos[idx])
            return (fd[:, :, None].repeat(1, 1, self.num_frames, 1), pd[:, :, None].repeat(1, 1, self.num_frames, 1))
        counts = m.sum(1).tolist() if _counts is None else _counts   # the one host round trip of this call
        rows = []
        for c in counts:                                          # generator calls as in the reference (:236-251)
            c = int(c)
            if c == 0:
                rows.append(torch.zeros(R, dtype=torch.int64))
            elif c < R:
                rows.append(torch.arange(c).repeat(int(R / c) + 1)[:R])
            else:
                assert not (prompt_type == "masks" and is_train), "training branch is out of scope"
                rows.append(torch.randperm(c)[:R])
        # one async transfer: the rank table plus a column flagging empty entities
        host = torch.cat([torch.stack(rows), torch.tensor([[int(int(c) == 0)] for c in counts], dtype=torch.int64)], dim=1)
        dev_tab = to_device_async(host, m.device)
        idx = _kth_true(m, dev_tab[:, :R])                        # [n, R] flat feature-map indices
        empty = (dev_tab[:, R] != 0).view(-1, 1, 1)
        self._last_dense = (idx, empty.view(-1))
        fd = torch.where(empty, query_feats[:, 0][:, None].expand(-1, R, -1), feats[idx])
        pd = torch.where(empty, query_pe[:, 0][:, None].expand(-1, R, -1), pos[idx])
        fd = fd[:, :, None].repeat(1, 1, self.num_frames, 1)
        pd = pd[:, :, None].repeat(1, 1, self.num_frames, 1)
        return fd, pd


class VisualPromptSampler:
    def __init__(self, pretrain_img_size=1024, hidden_dim=256, num_heads=8, num_frames=1, num_prev_frames_memory=1,
                 num_dense_points=32, position_embedding_sin3d_type="FixedT", clip_stride=1):
        self.num_heads = num_heads
        self.num_frames = num_frames
        self.key_fid = int((num_frames - 1) / 2)
        self.num_dense_points = num_dense_points
        self.clip_stride = clip_stride
        self.num_prev_frames_memory = max(num_prev_frames_memory, num_frames)
        self.visual_prompt_encoder = VisualPromptEncoder(pretrain_img_size, hidden_dim, num_frames, num_dense_points,
                                                         position_embedding_sin3d_type)
        self.prompt_feature_level_index = -1  # 1/8 resolution

    @torch.no_grad()
    def process_per_batch(self, img_emb_list, pos_emb_list, img_size_list, targets, training=True,
                          prompt_type="masks", use_all_prev_frames=False):
        if training:
            raise NotImplementedError("training-time prompt sampling is out of scope of the inference hot path")
        return self.process_per_batch_inference(img_emb_list, pos_emb_list, img_size_list, targets, prompt_type,
                                                use_all_prev_frames)

    @torch.no_grad()
    def process_per_batch_inference(self, img_emb_list, pos_emb_list, img_size_list, targets, prompt_type="masks",
                                    use_all_prev_frames=False):
        assert len(targets) == 1, "Only support batch size = 1 now"
        li = self.prompt_feature_level_index
        H, W = img_size_list[li]
        # '(H W) (N T) C -> N T C H W' with N = 1
        def to_ntchw(e):
            return e.view(H, W, 1, -1, e.shape[-1]).permute(2, 3, 4, 0, 1)
        img_emb, pos_emb = to_ntchw(img_emb_list[li]), to_ntchw(pos_emb_list[li])
        pe_l, f_l, m_l = [], [], []
        for ie, pe, tv in zip(img_emb, pos_emb, targets):
            tv["img_emb_per_video"] = ie
            tv["pos_emb_per_video"] = pe
            if "masks" not in tv or tv["masks"].nelement() == 0:
                return None, None, None
            o = self.process_per_video_inference(ie, pe, tv, prompt_type, False)
            pe_l.append(o[0]); f_l.append(o[1]); m_l.append(o[2])
        if len(f_l) == 0 or any(f is None for f in f_l):
            return None, None, None
        prompt_pe_dense = torch.stack(pe_l, dim=-3).flatten(-3, -2)       # n x R x NT x C
        prompt_feats_dense = torch.stack(f_l, dim=-3).flatten(-3, -2)
        prompt_attn_masks = torch.stack(m_l, dim=0).flatten(0, 1).repeat(1, self.num_heads, 1, 1).flatten(0, 1)
        # "avoid NaN" block, quirk kept (:836-840): multiplies by isblank, i.e. fills blanks with zeros
        isblank = (prompt_feats_dense == 0).all(-1)
        mean = (prompt_feats_dense * isblank.unsqueeze(-1)).flatten(1, 2).sum(1)
        mean = mean / isblank.flatten(1, 2).sum(1).unsqueeze(-1).clamp(min=1)
        mean = mean[:, None, None].repeat(1, prompt_feats_dense.shape[1], prompt_feats_dense.shape[2], 1)
        prompt_feats_dense = torch.where(isblank.unsqueeze(-1), mean, prompt_feats_dense)
        return prompt_pe_dense, prompt_feats_dense, prompt_attn_masks

    # ---- annotation-only work of a clip's `get_mask_prompt` calls, ahead of the image features ---------------------------
    def _needs_prev_frame(self, tv):
        return tv["first_frame_idx"] != 0 and ((self.num_frames == 1) or ("prompt_feats" not in tv))

    def _update_frames(self, tv, num_frames):
        # Important (reference comment): first clip encodes frame 0 only (none for grounding); later
        # clips re-encode all but the last `clip_stride` frames
        if tv["first_frame_idx"] == 0:
            return 1 - int(tv["task"] == "grounding")
        return num_frames - self.clip_stride

    @torch.no_grad()
    def _annotation_jobs(self, tv, num_frames, device, prompt_type="masks"):
        """Everything the `get_mask_prompt` calls of this clip compute from the annotations alone (candidate pixels,
        feature-resolution masks, the sizes of the random draws), for all key frames at once and on the CURRENT stream:
        {"prev": (ha, prefix, counts) | None, "clip": (prefix, counts) | None}; `counts` = per key frame the host list of
        draw sizes ("reference" sampler without replay: ONE host round trip for the whole clip) or None."""
        enc = self.visual_prompt_encoder
        cs, T = self.clip_stride, num_frames
        masks_all, boxes_all = tv["masks"], tv["boxes"]
        h, w = masks_all.shape[-2:]
        sc = enc.img_feats_scale
        h_img, w_img = h // sc, w // sc
        jobs = {"prev": None, "clip": None}
        if self._needs_prev_frame(tv):
            prev_frame_idx = max(0, tv["first_frame_idx"] - 1)
            fa = tv["first_appear_frame_idxs"]
            ha = torch.nonzero((fa <= prev_frame_idx) & (fa != -1)).flatten()      # the one extra host round trip
            if ha.numel() > 0:
                ha = ha.to(device)
                m = masks_all[:, -(T + cs):-T].to(device)[ha].transpose(0, 1)
                b = boxes_all[:, -(T + cs):-T].to(device)[ha].transpose(0, 1)
                jobs["prev"] = [ha, enc.annotation_prefix(m, b, h_img, w_img), None]
        U = self._update_frames(tv, T)
        if U > 0 and prompt_type == "masks":
            m = masks_all[:, -T:][:, :U].to(device).transpose(0, 1)
            b = boxes_all[:, -T:][:, :U].to(device).transpose(0, 1)
            jobs["clip"] = [enc.annotation_prefix(m, b, h_img, w_img), None]
        if enc._rng(device) == "reference" and enc._replay is None:
            parts = [j[-2]["counts"].flatten() for j in (jobs["prev"], jobs["clip"]) if j is not None]
            if parts:
                host = torch.cat(parts).tolist()                                   # ONE host round trip for the clip
                for j in (jobs["prev"], jobs["clip"]):
                    if j is not None:
                        Fk, n2 = j[-2]["counts"].shape
                        j[-1] = [host[k * n2:(k + 1) * n2] for k in range(Fk)]
                        host = host[Fk * n2:]
        return jobs

    @torch.no_grad()
    def prefetch(self, tv, num_frames):
        """Run `_annotation_jobs` for the NEXT `process_per_video_inference(..., tv)` on a side stream and hand its results
        over through `tv`.  Call it BEFORE the backbone of the clip is enqueued: the side stream then waits only for the
        work that produced the annotations (the previous clip), its kernels overlap the backbone, and the host round trip
        for the draw sizes waits for the side stream alone -- inside `get_mask_prompt` it would wait for the backbone and
        the pixel decoder, and the GPU would drain five times per clip (profiles/r03_prompted_clip_breakdown_v0.txt:
        21 ms of 55 idle).  Without this call the same work runs inline, once per clip."""
        if "masks" not in tv or tv["masks"].nelement() == 0 or not tv["masks"].is_cuda:
            return
        device = tv["masks"].device
        main = torch.cuda.current_stream(device)
        side = self.__dict__.setdefault("_side_streams", {}).get(str(device))
        if side is None:
            side = self._side_streams[str(device)] = torch.cuda.Stream(device=device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            jobs = self._annotation_jobs(tv, num_frames, device)
            ev = side.record_event()
        tv["_prompt_prefetch"] = {"jobs": jobs, "event": ev, "key": self._prefetch_key(tv, num_frames)}

    @staticmethod
    def _prefetch_key(tv, num_frames):
        """What the prefetched jobs were computed from: the clip position and the identity AND version of every annotation tensor
        they read -- an in-place update of masks / boxes / first-appearance indices between `prefetch` and the forward makes
        the key differ and the work is redone inline."""
        def ident(k):
            t = tv.get(k)
            return (id(t), t._version) if isinstance(t, torch.Tensor) else None
        return (tv["first_frame_idx"], num_frames, ident("masks"), ident("boxes"), ident("first_appear_frame_idxs"), ident("ids"))

    def _take_jobs(self, tv, num_frames, device, prompt_type):
        pf = tv.pop("_prompt_prefetch", None)
        if pf is not None and not (pf["key"] == self._prefetch_key(tv, num_frames) and prompt_type == "masks"):
            # stale or unused: the side stream's work is abandoned, but its tensors must not be freed under it
            torch.cuda.current_stream(device).wait_event(pf["event"])
            pf = None
        if pf is not None:
            main = torch.cuda.current_stream(device)
            main.wait_event(pf["event"])
            for j in (pf["jobs"]["prev"], pf["jobs"]["clip"]):
                if j is not None:
                    for t in j[-2].values():
                        t.record_stream(main)
                    if len(j) == 3:
                        j[0].record_stream(main)
            return pf["jobs"]
        return self._annotation_jobs(tv, num_frames, device, prompt_type)

    @torch.no_grad()
    def process_per_video_inference(self, img_emb, pos_emb, tv, prompt_type="masks", use_all_prev_frames=False):
        device = img_emb.device
        num_frames = img_emb.shape[0]
        first_frame_idx = tv["first_frame_idx"]
        frame_indices = tv["frame_indices"]
        is_first_clip = first_frame_idx == 0
        jobs = self._take_jobs(tv, num_frames, device, prompt_type)
        if not is_first_clip:
            self.zero_pad_prompt(tv)
            self.process_per_video_inference_prev_frame(tv, prompt_type="masks", _job=jobs["prev"])
        gt_boxes = tv["boxes"][:, -num_frames:].to(device)
        gt_masks = tv["masks"][:, -num_frames:].to(device)
        update_frames = self._update_frames(tv, num_frames)
        enc = self.visual_prompt_encoder
        assert prompt_type in {"boxes", "masks"}, "point prompts at inference are not supported"
        batched = None
        if prompt_type == "masks" and update_frames > 0:
            pre, counts = jobs["clip"]
            if enc.feature_reduce is None:
                # all key frames of the clip in one pass (same draws, in the same order, as one call per frame)
                U = update_frames
                batched = enc.get_mask_prompts(img_emb[:U], pos_emb[:U], gt_masks[:, :U].transpose(0, 1),
                                               gt_boxes[:, :U].transpose(0, 1), list(range(U)),
                                               [frame_indices[k] for k in range(U)], pre, counts)
        for key_fid in range(update_frames):
            kfo = frame_indices[key_fid]
            x_key, x_pos = img_emb[key_fid], pos_emb[key_fid]
            if prompt_type == "boxes":
                tup = enc.get_box_prompt(x_key, x_pos, gt_boxes[:, key_fid], is_train=False, key_fid=key_fid,
                                         key_fid_original=kfo)
            elif batched is not None:
                tup = tuple(b[key_fid] for b in batched)
            else:
                tup = enc.get_mask_prompt(x_key, x_pos, masks=gt_masks[:, key_fid], boxes=gt_boxes[:, key_fid],
                                          is_train=False, key_fid=key_fid, key_fid_original=kfo,
                                          _pre={k: v[key_fid] for k, v in pre.items()},
                                          _counts=None if counts is None else counts[key_fid])
            pe_d, f_d, m_d = tup[1], tup[2], tup[3]
            tv["prompt_obj_ids"] = tv["ids"]
            if is_first_clip:
                tv["prompt_pe"], tv["prompt_feats"], tv["prompt_attn_masks"] = pe_d, f_d, m_d
            else:
                s_idx = -num_frames + key_fid
                # entities visible in this frame overwrite their pool rows (select, not boolean-mask indexing:
                # that would cost a host round trip per frame)
                valid = pre["visible"][key_fid].view(-1, 1, 1, 1) if prompt_type == "masks" else \
                    (gt_masks[:, key_fid].amax(2).amax(1) > 0).view(-1, 1, 1, 1)
                for name, new in (("prompt_pe", pe_d), ("prompt_feats", f_d)):
                    dst = tv[name][:, :, s_idx:]
                    torch.where(valid, new[:, :, key_fid:], dst, out=dst)     # in place: one launch, no temporary + copy
                tv["prompt_attn_masks"][s_idx:] = m_d[key_fid:]
        if "prompt_pe" not in tv:
            return None, None, None
        return (tv["prompt_pe"][:, :, -num_frames:], tv["prompt_feats"][:, :, -num_frames:],
                tv["prompt_attn_masks"][-num_frames:])

    @torch.no_grad()
    def process_per_video_inference_prev_frame(self, tv, prompt_type="masks", _job=None):
        device = tv["img_emb_per_video"].device
        n_inst = tv["masks"].shape[0]
        num_frames = tv["img_emb_per_video"].shape[0]
        if not self._needs_prev_frame(tv):
            return
        if _job is None:
            _job = self._annotation_jobs(tv, num_frames, device)["prev"]
        if _job is None:                                          # no entity has appeared yet
            return
        ha, pre, counts = _job
        cs = self.clip_stride
        enc = self.visual_prompt_encoder
        assert prompt_type == "masks"
        batched = None
        if enc.feature_reduce is None:
            T = num_frames
            batched = enc.get_mask_prompts(tv["img_emb_per_video"][:cs], tv["pos_emb_per_video"][:cs],
                                           tv["masks"][:, -(T + cs):-T].to(device)[ha].transpose(0, 1),
                                           tv["boxes"][:, -(T + cs):-T].to(device)[ha].transpose(0, 1), list(range(cs)),
                                           [tv["frame_indices"][0] - (cs - k) for k in range(cs)], pre, counts)
        for key_fid in range(cs):
            if batched is not None:
                tup = tuple(b[key_fid] for b in batched)
            else:
                gt_boxes = tv["boxes"][:, -(num_frames + cs) + key_fid].to(device)[ha]
                gt_masks = tv["masks"][:, -(num_frames + cs) + key_fid].to(device)[ha]
                kfo = tv["frame_indices"][0] - (cs - key_fid)
                x_key, x_pos = tv["img_emb_per_video"][key_fid], tv["pos_emb_per_video"][key_fid]
                tup = enc.get_mask_prompt(x_key, x_pos, masks=gt_masks, boxes=gt_boxes,
                                          is_train=False, key_fid=key_fid, key_fid_original=kfo,
                                          _pre={k: v[key_fid] for k, v in pre.items()},
                                          _counts=None if counts is None else counts[key_fid])
            pe_d, f_d, m_d = tup[1], tup[2], tup[3]
            if "prompt_feats" not in tv:
                _, R, T, C = pe_d.shape
                tv["prompt_pe"] = torch.zeros([n_inst, R, T + cs, C], device=device)
                tv["prompt_feats"] = torch.zeros([n_inst, R, T + cs, C], device=device)
                tv["prompt_attn_masks"] = torch.zeros([T + cs, m_d.shape[1], n_inst, m_d.shape[-1]], device=device).bool()
            col = -(num_frames + cs) + key_fid
            tv["prompt_pe"][ha, :, col] = pe_d[:, :, key_fid]
            tv["prompt_feats"][ha, :, col] = f_d[:, :, key_fid]
            tv["prompt_attn_masks"][col, :, ha] = m_d[key_fid]

    @torch.no_grad()
    def zero_pad_prompt(self, tv):
        if "prompt_feats" not in tv:
            return
        cs = self.clip_stride
        z = torch.zeros_like(tv["prompt_pe"][:, :, -cs:])
        tv["prompt_pe"] = torch.cat([tv["prompt_pe"], z], dim=2)
        tv["prompt_feats"] = torch.cat([tv["prompt_feats"], z], dim=2)
        tv["prompt_attn_masks"] = torch.cat([tv["prompt_attn_masks"], tv["prompt_attn_masks"][-cs:]])
        tv["prompt_attn_masks"][-cs:] = False
