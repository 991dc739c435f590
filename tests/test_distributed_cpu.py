"""CPU, world_size 2, gloo: the N>1 paths.
  * frame-sharded decoder (univs_amd/distributed.py): a clip's frames split over 2 ranks, one all-gather of the
    query states per decoder layer, all-reduced means over T -- must reproduce the single-process result;
  * the bench's clip-replica mode needs no collective on the data path; only its timing reduction is exercised.
The four HIP operators are replaced by the oracle's CPU stand-ins (oracle/cpu_path.py), as in test_modules_cpu."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, scenario):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from oracle.cpu_path import cpu_ops
        from tests import cases, helpers
        from univs_amd.distributed import FrameShard, shard_frames

        case = dict(cases.HEAD_CASE, name="head_dist", T=4)
        feats = cases.backbone_features(case)
        if scenario == "visual_prompts":
            return _visual_prompt_clips(case, feats, world)
        if scenario == "first_clip":
            dec_over, targets_fn = {}, lambda: cases.targets_first_clip(case)
        else:
            dec_over = dict(text_to_image=True, sa_mask="sep-blocked")
            targets_fn = lambda: cases.targets_grounding(case)  # noqa: E731
        head = helpers.build_head(case, "cpu", return_aux=False, **dec_over)
        with cpu_ops(), torch.no_grad():
            ref = head(feats, targets=targets_fn())                       # single-process result (all 4 frames)
            shard = FrameShard()
            head.predictor.frame_shard = shard
            out = head(shard_frames(feats, shard, case["T"]), targets=targets_fn())
        sl = shard.local_slice(case["T"] // world)
        assert out["pred_masks"].shape[2] == case["T"] // world
        err_m = (out["pred_masks"] - ref["pred_masks"][:, :, sl]).abs().max().item()
        err_l = (out["pred_logits"] - ref["pred_logits"]).abs().max().item()
        err_e = (out["pred_embds"] - ref["pred_embds"][:, :, sl]).abs().max().item()
        assert err_m < 2e-4 and err_l < 2e-4 and err_e < 2e-4, (rank, err_m, err_l, err_e)
        flips = ((out["pred_masks"] > 0) != (ref["pred_masks"][:, :, sl] > 0))[ref["pred_masks"][:, :, sl].abs() > 1e-3]
        assert flips.sum() == 0
        # bench-style timing reduction: max over ranks
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == float(world)
    finally:
        dist.destroy_process_group()


def _visual_prompt_clips(case, feats, world):
    """Second and third clip of a video with three tracked entities: prompt sampler + memory pool + ProCA, frames sharded
    over the ranks (sampler replicated, token features summed over ranks) == the single-process result, pool included."""
    from oracle.cpu_path import cpu_ops
    from tests import cases, helpers
    from univs_amd.distributed import FrameShard, shard_frames
    T = case["T"]

    def advance(tv):             # what the clip loop does between two clips at stride 1 (cf. helpers.advance_to_third_clip)
        tv["first_frame_idx"] = 2
        tv["frame_indices"] = torch.arange(2, 2 + T)
        for k in ("masks", "boxes"):
            tv[k] = torch.cat([tv[k], torch.zeros_like(tv[k][:, :1])], 1)
            tv[k][:, -2] = tv[k][:, -3]
        tv["ids"] = torch.cat([tv["ids"], tv["ids"][:, :1]], 1)

    def two_clips(head, f):
        targets = cases.targets_with_entities(case, first_frame_idx=1, n_ent=3)
        outs = []
        for clip in range(2):
            torch.manual_seed(clip)          # the sampler draws from the CPU generator: same state on every rank
            outs.append(head(f, targets=targets))
            if clip == 0:
                advance(targets[0])
        return outs, targets[0]

    head = helpers.build_head(case, "cpu", return_aux=False)
    with cpu_ops(), torch.no_grad():
        ref, tv_ref = two_clips(head, feats)
        shard = FrameShard()
        head.predictor.frame_shard = shard
        out, tv = two_clips(head, shard_frames(feats, shard, T))
    sl = shard.local_slice(T // world)
    assert ref[0]["pred_masks"].shape[1] == case["Q"] + 3          # learnable + one prompt query per entity
    for c in range(2):
        assert out[c]["pred_masks"].shape == ref[c]["pred_masks"][:, :, sl].shape
        err_m = (out[c]["pred_masks"] - ref[c]["pred_masks"][:, :, sl]).abs().max().item()
        err_l = (out[c]["pred_logits"] - ref[c]["pred_logits"]).abs().max().item()
        err_e = (out[c]["pred_embds"] - ref[c]["pred_embds"][:, :, sl]).abs().max().item()
        assert err_m < 2e-4 and err_l < 2e-4 and err_e < 2e-4, (c, err_m, err_l, err_e)
    # the memory pool is replicated: every rank ends up with the single-process pool
    for k in ("prompt_feats", "prompt_pe"):
        assert tv[k].shape == tv_ref[k].shape and (tv[k] - tv_ref[k]).abs().max().item() < 1e-5, k
        assert tv[k].abs().max().item() > 0
    assert torch.equal(tv["prompt_attn_masks"], tv_ref["prompt_attn_masks"])
    # only this rank's frames of the stored features are real, the rest are zeros (they never cross the fabric)
    other = [t for t in range(T) if not (sl.start <= t < sl.stop)]
    assert tv["img_emb_per_video"][other].abs().max().item() == 0
    assert torch.equal(tv["img_emb_per_video"][sl], tv_ref["img_emb_per_video"][sl])


def _box_point_worker(rank, world, port):
    """Box and point prompts in frame-sharded mode (prompt_encoder.py:362-442): the rank that owns the key frame holds its
    features, the others zeros; token features summed over the ranks == the single-process tokens on every rank."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from univs_amd import synth
        from univs_amd.distributed import FrameShard
        from univs_amd.modeling.prompt_encoder import VisualPromptEncoder
        enc = VisualPromptEncoder(pretrain_img_size=64, hidden_dim=32, num_frames=4, num_dense_points=6,
                                  position_embedding_sin3d_type="FixedT")
        C, h, w = 32, 8, 12
        feats = synth.normal("dist/bp/f", (C, h, w))
        pos = synth.normal("dist/bp/p", (C, h, w))
        boxes = torch.tensor([[0.1, 0.2, 0.6, 0.7], [0.5, 0.5, 0.9, 0.95], [0.3, 0.3, 0.3, 0.3], [0.92, 0.93, 0.99, 0.99]])
        points = torch.tensor([[0.2, 0.3], [0.8, 0.6], [1.2, 0.5]])

        def run():
            torch.manual_seed(3)          # the dense-token sampler draws from the CPU generator: same state on every rank
            b = enc.get_box_prompt(f_local, pos, boxes)
            torch.manual_seed(4)
            p = enc.get_point_prompt(f_local, pos, point_coords=points)
            return b, p

        f_local = feats
        ref_b, ref_p = run()
        shard = FrameShard()
        f_local = feats if rank == 1 else torch.zeros_like(feats)      # rank 1 owns the key frame
        enc.feature_reduce = shard.all_reduce_sum
        got_b, got_p = run()
        for got, ref in ((got_b, ref_b), (got_p, ref_p)):
            for g_, r_ in zip(got, ref):
                assert g_.shape == r_.shape
                if g_.dtype == torch.bool:
                    assert torch.equal(g_, r_)
                else:
                    assert (g_ - r_).abs().max().item() < 1e-6
        assert ref_b[2].abs().max().item() > 0 and ref_p[2].abs().max().item() > 0
    finally:
        dist.destroy_process_group()


def test_frame_sharded_box_and_point_prompts():
    world = 2
    mp.spawn(_box_point_worker, args=(world, _free_port()), nprocs=world, join=True)


@pytest.mark.parametrize("scenario", ["first_clip", "grounding", "visual_prompts"])
def test_frame_sharded_decoder_matches_single_process(scenario):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), scenario), nprocs=world, join=True)


def test_one_rank_group_with_collectives_issued():
    """bench.py's `frame_sharded_n1`: FrameShard(always_collective=True) in a ONE-rank group runs the collectives for real
    and must reproduce the unsharded result exactly (a gather of one block and a sum of one term are identities)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from oracle.cpu_path import cpu_ops
        from tests import cases, helpers
        from univs_amd.distributed import FrameShard

        shard = FrameShard(always_collective=True)
        x = torch.arange(24.0).view(2, 3, 4)
        for dim in (0, 1, -1):
            y = shard.all_gather_frames(x, dim)
            assert y.data_ptr() != x.data_ptr() and torch.equal(y, x)
        assert torch.equal(shard.all_reduce_sum(x.clone()), x)
        assert FrameShard().all_gather_frames(x, 1) is x          # default: a one-rank shard returns its input

        case = dict(cases.HEAD_CASE, name="head_dist1", T=4)
        feats = cases.backbone_features(case)
        head = helpers.build_head(case, "cpu", return_aux=False)
        with cpu_ops(), torch.no_grad():
            ref = head(feats, targets=cases.targets_first_clip(case))
            head.predictor.frame_shard = shard
            out = head(feats, targets=cases.targets_first_clip(case))
        for k in ("pred_masks", "pred_logits", "pred_embds"):
            assert torch.equal(out[k], ref[k]), k
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pre_norm", [False, True], ids=["post_norm", "pre_norm"])
def test_self_attention_rows_equal_rows_of_the_full_attention(pre_norm):
    """The row-sharded form of the spatio-temporal self-attention (…decoder_univs.py:408-414 evaluated per rank only for the
    query rows of its own frames): rows [Q' T_loc] against all [Q' T] keys == the same rows of the full
    attention, with the 'sep-blocked' mask sliced by rows; and the layer really only projects / attends Q' T_loc query rows."""
    from oracle.cpu_path import cpu_ops
    from univs_amd import synth
    from univs_amd.modeling.transformer_decoder.transformer_layers import SelfAttentionLayer
    Qn, T, C, nq = 7, 6, 32, 4
    layer = SelfAttentionLayer(C, 4, normalize_before=pre_norm).eval()
    with torch.no_grad():
        for n, p in layer.named_parameters():
            p.copy_(synth.normal(f"sa_rows/{n}", tuple(p.shape), std=0.2))
    x = synth.normal("sa_rows/x", (Qn, T, C))
    pos = synth.normal("sa_rows/pos", (Qn, T, C))
    mask = torch.ones(Qn * T, Qn * T, dtype=torch.bool)
    mask[:nq * T, :nq * T] = False
    for j in range(Qn - nq):
        s = (nq + j) * T
        mask[s:s + T, s:s + T] = False
    seen = []
    attn_forward = layer.self_attn.forward
    layer.self_attn.forward = lambda q, k, v, **kw: (seen.append((q.shape[0], k.shape[0])), attn_forward(q, k, v, **kw))[1]
    with cpu_ops(), torch.no_grad():
        full = layer(x.reshape(Qn * T, 1, C), tgt_mask=mask, query_pos=pos.reshape(Qn * T, 1, C)).reshape(Qn, T, C)
        for lo, hi in ((0, 2), (2, 4), (4, 6), (1, 2)):
            sl = slice(lo, hi)
            rows_mask = mask.view(Qn, T, Qn * T)[:, sl].reshape(-1, Qn * T)
            got = layer(x[:, sl].reshape(-1, 1, C), tgt_mask=rows_mask, query_pos=pos[:, sl].reshape(-1, 1, C),
                        kv=x.reshape(Qn * T, 1, C), kv_pos=pos.reshape(Qn * T, 1, C)).reshape(Qn, hi - lo, C)
            assert (got - full[:, sl]).abs().max().item() < 2e-6
            assert seen[-1] == (Qn * (hi - lo), Qn * T)
    assert seen[0] == (Qn * T, Qn * T)


# ---------------------------------------------------------------------------------------------------
# the reference's sliding clip loop, frames of the video spread over the ranks (inference/video_entity.py: set_frame_shard)
# ---------------------------------------------------------------------------------------------------
LOOP_STATE_KEYS = ("logits", "masks", "mask_logits", "boxes", "embds", "ids", "first_appear_frame_idxs", "mask_quality_scores",
                   "occurrence", "prompt_pe", "prompt_feats", "prompt_attn_masks", "frame_indices")


def _loop_states(case, model, shard, seed=1, replicate_clip_masks=False, **over):
    """Runs the clip loop (sharded if `shard`); returns ({tag_key: tensor}, results, clip starts, pixel-decoder calls)."""
    from tests import cases
    from univs_amd.inference.video_entity import ImageList, InferenceVideoEntity
    inf = InferenceVideoEntity(**cases.loop_kwargs(case, **over))
    inf.set_frame_shard(shard)
    inf.replicate_clip_masks = replicate_clip_masks      # True: every rank receives every query's mask logits (the form before ClipMaskRows)
    if shard is None:
        inf.pixel_decoder_once_per_window = False      # the comparison run: the reference's call pattern (pixel decoder per clip)
    dumps, calls = {}, []

    def snapshot(tag, tv):
        for k in LOOP_STATE_KEYS:
            if k in tv:
                dumps[f"{tag}_{k}"] = tv[k].detach().float().clone() if tv[k].dtype == torch.bool else tv[k].detach().clone()

    def on_clip(i, targets):
        snapshot(f"clip{len(calls)}_in", targets[0])
        calls.append(int(i))
    pd = model.sem_seg_head.pixel_decoder
    pd_frames = []
    orig = pd.forward_features
    pd.forward_features = lambda f: (pd_frames.append(int(next(iter(f.values())).shape[0])), orig(f))[1]
    try:
        x = cases.preprocess(cases.loop_frames(case))
        images = ImageList(x, [case["image_size"]] * case["n_frames"])
        targets = cases.loop_targets(case)
        with torch.no_grad():
            torch.manual_seed(seed)         # the prompt sampler draws from the CPU generator (the sharded loop installs rank 0's state)
            results = inf.inference_video(model, cases.loop_batched_inputs(case), images, targets, merge_results=False, on_clip=on_clip)
    finally:
        pd.forward_features = orig
    snapshot("final", targets[0])
    return dumps, results, calls, pd_frames


def _clip_loop_worker(rank, world, port, case_over, kw_over):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        import types
        from oracle.cpu_path import cpu_ops
        from tests import cases, helpers
        from univs_amd.distributed import FrameShard
        case = dict(cases.LOOP_CASE, **case_over)
        model = types.SimpleNamespace(backbone=helpers.build_swin("cpu"), sem_seg_head=helpers.build_head(case, "cpu"))
        kw = dict(stability_score_thresh=0.0, **kw_over)
        with cpu_ops():
            ref, ref_results, ref_calls, ref_pd = _loop_states(case, model, None, **kw)
            # rank-dependent seeds (detectron2's seed + rank): the loop broadcasts rank 0's generator state at the start of the video
            shard = FrameShard()
            got, results, calls, pd_frames = _loop_states(case, model, shard, seed=1 + 1000 * rank, **kw)
        assert calls == ref_calls and len(calls) >= 2
        if world == 2 and case_over:
            # the clip's mask logits stay sharded (ClipMaskRows: statistics of all rows, the logits of the few rows that enter the
            # per-video state, the candidates' IoU by a maximum): fewer result bytes than replicating every row, the same states
            with cpu_ops():
                shard_r = FrameShard()
                got_r, _, _, _ = _loop_states(case, model, shard_r, seed=1 + 1000 * rank, replicate_clip_masks=True, **kw)
            lazy_b = shard.bytes["result:all_gather"] + shard.bytes["result:all_reduce"]
            print(f"rank {rank}/{world}: result bytes per clip {lazy_b // len(calls)} with sharded mask logits, {shard_r.bytes['result:all_gather'] // len(calls)} replicated")
            assert 0 < lazy_b < 0.6 * shard_r.bytes["result:all_gather"], (lazy_b, dict(shard_r.bytes))
            assert sorted(got_r) == sorted(got) and all(torch.equal(got_r[k], got[k]) for k in got), [k for k in got if not torch.equal(got_r[k], got[k])]
        # what crossed the ranks (bytes received by this rank): the decoder's own collectives -- one all-gather of the query states per
        # layer, the sums of the sampled prompt tokens -- stay small; the bulk is the replication of the clip's mask logits for the
        # book-keeping that every rank repeats (DESIGN.md section 6 states it; SURVEY 8e's sharded post-processing is not built)
        per_clip = {k: v / len(calls) for k, v in shard.bytes.items()}
        print(f"rank {rank}/{world}: bytes received per clip {dict((k, int(v)) for k, v in per_clip.items())}")
        assert 0 < per_clip["all_gather"] + per_clip["all_reduce"] <= 4 * 2 ** 20, per_clip
        assert per_clip["result:all_gather"] > 0 or per_clip["result:broadcast"] > 0
        if world > case["T"]:
            assert shard.bytes["result:broadcast"] > 0 and len(shard._subgroups) >= 2     # this rank sat out some clip; several teams
        # the stateless part really is spread: this rank ran backbone + pixel decoder on ITS frames only, once per frame ...
        # (windows as the reference cuts them, :309-312: a new window starts with the first clip that reaches past the last one)
        n, T, W = case["n_frames"], case["T"], 5
        expect, win_end = [], 0
        for i in ref_calls:
            if i + T > win_end:
                win_end = i + W
                expect.append(len([f for f in range(i, min(win_end, n)) if f % world == rank]))
        assert pd_frames == [e for e in expect if e], (pd_frames, expect, rank)
        # ... where the unsharded loop runs the pixel decoder on every clip's frames
        assert sum(ref_pd) == sum(min(case["T"], n - i) for i in ref_calls)
        assert sorted(got) == sorted(ref)
        for k in sorted(ref):
            a, b = got[k], ref[k]
            assert a.shape == b.shape, (k, a.shape, b.shape)
            if a.numel() == 0:
                continue
            if k.endswith("_masks") and not k.endswith("attn_masks"):
                logit = ref[k.replace("_masks", "_mask_logits")]
                sure = logit.abs() > 1e-3                  # binarised logits: away from the decision boundary
                assert torch.equal(a[sure], b[sure]), k
            elif not a.dtype.is_floating_point:
                assert torch.equal(a, b), k
            else:
                err = (a.double() - b.double()).abs().max().item()
                assert err <= 2e-4 * max(1.0, b.abs().max().item()), (k, err)
        assert len(results) == len(ref_results)
        for ra, rb in zip(results, ref_results):
            assert [r["obj_id"] for r in ra] == [r["obj_id"] for r in rb]
            for x_, y_ in zip(ra, rb):
                assert (torch.as_tensor(x_["score"]).double() - torch.as_tensor(y_["score"]).double()).abs().max().item() < 1e-4
                assert (x_["masks"] != y_["masks"]).float().mean().item() < 1e-3
        # every rank holds the same (replicated) state: compare a digest across the ranks
        digest = torch.stack([got[k].double().sum() for k in sorted(got) if got[k].dtype.is_floating_point and got[k].numel()])
        both = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(both, digest)
        assert all((d - both[0]).abs().max().item() <= 1e-6 * max(1.0, both[0].abs().max().item()) for d in both)
    finally:
        dist.destroy_process_group()


def test_sharded_clip_loop_with_more_ranks_than_frames():
    """FOUR ranks on 3-frame clips (the loop used to refuse a group larger than a clip -- 5 of a node's 8 GPUs on the
    reference's 5-frame clips): frame f on rank f % 4, every clip's decoder on the three ranks that own one of its frames (a sub-group
    per distinct team), its outputs / the prompt pool / the generator states handed to the fourth.  The per-video state at the entry
    of every clip and at the end and the emitted results == the single-process loop, on every rank; every rank ran backbone + pixel
    decoder on its own frames only."""
    world = 4
    mp.spawn(_clip_loop_worker, args=(world, _free_port(), dict(n_frames=8), dict(clip_stride=1)), nprocs=world, join=True)


@pytest.mark.parametrize("case_over,kw_over", [({}, {}), (dict(n_frames=8), dict(clip_stride=1))], ids=["g11a-7frames-stride2", "8frames-stride1"])
def test_sharded_clip_loop_matches_single_process(case_over, kw_over):
    """The reference's sliding clip loop (inference_video_entity.py:296-316; the 7-frame video of golden g11a) with the frames spread
    over 2 ranks (frame f on rank f % 2: non-contiguous, unequal shares of every 3-frame clip): the per-video state at the entry of
    every clip and at the end, and the emitted results == the single-process loop on every rank; backbone + pixel decoder run once per
    owned frame and window.  Stride 1 makes the clips overlap by two frames."""
    world = 2
    mp.spawn(_clip_loop_worker, args=(world, _free_port(), case_over, kw_over), nprocs=world, join=True)


def test_clip_shard_gather_orders_frames():
    """ClipShard.all_gather_frames in a one-rank group (collective issued): identity; the position bookkeeping for cyclic owners."""
    from univs_amd.distributed import ClipShard, cyclic_owners
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        assert cyclic_owners(3, 5, 2) == [1, 0, 1, 0, 1] and cyclic_owners(4, 3, 8) == [4, 5, 6]
        cs = ClipShard([0, 0, 0], always_collective=True)
        x = torch.arange(24.0).view(2, 3, 4)
        assert torch.equal(cs.all_gather_frames(x, 1), x) and cs.local_slice(3) == slice(0, 3) and cs.total(3) == 3
        with pytest.raises(AssertionError):
            ClipShard([0, 1, 0])               # rank 1 does not exist in a one-rank group
    finally:
        dist.destroy_process_group()


def _vos_loop_worker(rank, world, port):
    """The config-built model (tests/test_meta_arch_cpu.py) on a mask-prompted ('sot') request: the VOS driver with the frames spread over
    the ranks == the single-process driver (id maps of every frame), on every rank."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from oracle.cpu_path import cpu_ops
        from tests.test_meta_arch_cpu import H, N_FRAMES, W, make_model, video
        from univs_amd.distributed import FrameShard
        from univs_amd.inference.video_vos import FrameAnnotations
        model = make_model()
        m = torch.zeros(1, H, W)
        m[0, 10:40, 20:60] = 1
        ann0 = FrameAnnotations((H, W), [7], m, torch.tensor([[20.0, 10.0, 60.0, 40.0]]), torch.tensor([0]))

        def request():
            anns = [ann0] + [FrameAnnotations((H, W)) for _ in range(N_FRAMES - 1)]
            return video("sot", "ytbvos18_val", instances=anns, mask_palette=[0] * 768)
        pd = model.sem_seg_head.pixel_decoder
        frames_seen = []
        orig = pd.forward_features
        pd.forward_features = lambda f: (frames_seen.append(int(next(iter(f.values())).shape[0])), orig(f))[1]
        with cpu_ops():
            torch.manual_seed(0)
            ref = torch.cat(model(request()))
            n_ref = sum(frames_seen)
            frames_seen.clear()
            model.inference_video_vos.set_frame_shard(FrameShard())
            torch.manual_seed(0)
            got = torch.cat(model(request()))
            model.inference_video_vos.set_frame_shard(None)
        assert tuple(got.shape) == (N_FRAMES, H, W) and got.dtype == torch.uint8
        assert (got != ref).float().mean().item() < 2e-3, (got != ref).float().mean().item()     # (pixels at a logit of ~0 may differ)
        assert set(got.unique().tolist()) <= {0, 7} and torch.equal(got[0] == 7, m[0].bool())
        owned = len([f for f in range(N_FRAMES) if f % world == rank])
        assert owned <= sum(frames_seen) < n_ref, (frames_seen, n_ref)     # this rank ran the pixel decoder on its own frames only
        both = [torch.zeros_like(got) for _ in range(world)]
        dist.all_gather(both, got)
        assert all(torch.equal(b, both[0]) for b in both)                     # the ranks agree exactly
    finally:
        dist.destroy_process_group()


def test_sharded_vos_driver_matches_single_process():
    """InferenceVideoVOS.set_frame_shard (inference_video_vos.py:243-284 with frame f on rank f % 2): backbone + pixel decoder on the owned
    frames, every clip's decoder on both ranks (ClipShard), per-video state replicated; the id maps equal the single-process driver's."""
    world = 2
    mp.spawn(_vos_loop_worker, args=(world, _free_port()), nprocs=world, join=True)
