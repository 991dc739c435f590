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
