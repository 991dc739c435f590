"""CPU: the config-built inference shell (univs_amd/modeling/meta_arch/univs_prompt.py: UniVS_Prompt) end to end on a tiny
video -- batched_inputs as the reference's dataset mapper hands them over -> PrepareTargets -> clip loop -> results -- for the
three request kinds of `forward_inference` (univs/univs_prompt.py:416-452): category-specified (detection), mask-prompted
('sot') and text-prompted (custom expressions through tokenizer + CLIP text tower).  Operators: the oracle's CPU stand-ins."""
import pytest
import torch

from oracle.cpu_path import cpu_ops
from tests import cases
from tests.test_language_cpu import BPE, needs_vocab
from univs_amd import synth
from univs_amd.config import get_cfg
from univs_amd.inference.video_vos import FrameAnnotations
from univs_amd.modeling.build import build_model
from univs_amd.modeling.prompt_encoder import TextPromptEncoder

N_FRAMES, H, W = 5, 60, 90


def make_model(**test_over):
    cfg = get_cfg()
    cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES = 20
    cfg.INPUT.SAMPLING_FRAME_NUM = 3
    cfg.MODEL.BoxVIS.TEST.CLIP_STRIDE = 2
    cfg.MODEL.UniVS.CLIP_CLASS_EMBED_PATH = cases.clip_table()
    cfg.MODEL.UniVS.TEST.VIDEO_UNIFIED_INFERENCE_ENABLE = True
    cfg.MODEL.UniVS.TEST.CLIP_STRIDE = 2
    for k, v in test_over.items():
        cfg.MODEL.UniVS.TEST[k] = v
    model = build_model(cfg).eval()
    synth.load_synthetic(model)
    return model


def video(task, dataset, **extra):
    frames = synth.synthetic_frames(N_FRAMES, H, W, "meta/frames")
    d = {"image": [f for f in frames], "video_len": N_FRAMES, "height": H, "width": W, "task": task, "dataset_name": dataset,
         "file_names": [f"videos/v0/{i:05d}.jpg" for i in range(N_FRAMES)], "is_raw_video": False}
    d.update(extra)
    return [d]


def test_detection_request_runs_the_entity_loop():
    model = make_model()
    calls = []
    # (the loop calls the head's predictor per clip; the pixel decoder runs once for the 5-frame window: inference/video_entity.py)
    pred = model.sem_seg_head.predictor.forward
    model.sem_seg_head.predictor.forward = lambda *a, **k: (calls.append((a[4] if len(a) > 4 else k["targets"])[0]["first_frame_idx"]), pred(*a, **k))[1]
    with cpu_ops():
        torch.manual_seed(0)
        out = model(video("detection", "ytvis21", video_id=3))
    assert calls == [0, 2]                                        # 5 frames, clips of 3, stride 2: the second clip is the last
    assert isinstance(out, list)
    for r in out:                                                 # per-video YouTube-VIS records, json-serialisable RLEs
        assert r["video_id"] == 3 and len(r["segmentations"]) == N_FRAMES and r["height"] == H and r["width"] == W
        assert 0 <= r["category_id"] < 40 and isinstance(r["segmentations"][0]["counts"], str)


def test_sot_request_runs_the_vos_loop():
    model = make_model()
    m = torch.zeros(1, H, W)
    m[0, 10:40, 20:60] = 1
    ann0 = FrameAnnotations((H, W), [7], m, torch.tensor([[20.0, 10.0, 60.0, 40.0]]), torch.tensor([0]))
    anns = [ann0] + [FrameAnnotations((H, W)) for _ in range(N_FRAMES - 1)]
    with cpu_ops():
        torch.manual_seed(0)
        out = model(video("sot", "ytbvos18_val", instances=anns, mask_palette=[0] * 768))
    idmaps = torch.cat(out)
    assert tuple(idmaps.shape) == (N_FRAMES, H, W) and idmaps.dtype == torch.uint8
    assert set(idmaps.unique().tolist()) <= {0, 7}
    assert torch.equal(idmaps[0] == 7, m[0].bool())              # the annotated frame keeps its annotation


@needs_vocab
def test_custom_text_request_goes_from_raw_text_to_masks(monkeypatch):
    monkeypatch.setenv("UNIVS_BPE_VOCAB", BPE)
    from univs_amd.modeling.language import tokenizer as tk
    tk._shared_tokenizer.cache_clear()
    model = make_model(CUSTOM_VIDEOS_TEXT=[["a dog running", "the person on the left"]], CUSTOM_VIDEOS_ENABLE=True)
    enc = cases.build_text_encoder(dict(cases.TEXT_SMALL, embed_dim=640))
    model.text_prompt_encoder = TextPromptEncoder(enc, num_frames=3, device="cpu")
    with cpu_ops():
        torch.manual_seed(0)
        out = model(video("detection", "my_videos", is_raw_video=True))
    assert all(r["ids"] == [0, 1] for r in out)
    masks = torch.cat([r["masks"] for r in out], dim=1)           # [expressions, frames, H, W]
    assert tuple(masks.shape) == (2, N_FRAMES, H, W) and set(masks.unique().tolist()) <= {0, 255}


def test_unbuilt_branches_are_loud():
    model = make_model(VIDEO_UNIFIED_INFERENCE_ENABLE=False)
    with pytest.raises(NotImplementedError):
        model(video("detection", "ytvis_2021"))
    with pytest.raises(NotImplementedError):
        model(video("detection", "coco_2017_val"))
    model.train()
    with pytest.raises(NotImplementedError):
        model(video("detection", "ytvis_2021"))


def test_longvideo_shell_shares_the_inference_dispatch():
    from univs_amd.registry import META_ARCH_REGISTRY
    from univs_amd.modeling.meta_arch.univs_prompt import UniVS_Prompt, UniVS_Prompt_LongVideo
    assert META_ARCH_REGISTRY.get("UniVS_Prompt") is UniVS_Prompt
    assert META_ARCH_REGISTRY.get("UniVS_Prompt_LongVideo") is UniVS_Prompt_LongVideo
    cfg = get_cfg()
    cfg.MODEL.META_ARCHITECTURE = "UniVS_Prompt_LongVideo"
    cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES = 20
    cfg.INPUT.SAMPLING_FRAME_NUM = 3
    cfg.MODEL.UniVS.CLIP_CLASS_EMBED_PATH = cases.clip_table()
    cfg.MODEL.UniVS.TEST.VIDEO_UNIFIED_INFERENCE_ENABLE = True
    model = build_model(cfg).eval()
    assert isinstance(model, UniVS_Prompt_LongVideo)
    seen = []
    model.inference_video_entity.eval = lambda m, b: seen.append("entity") or []
    model.inference_video_vos.eval = lambda m, b: seen.append("vos") or []
    model(video("detection", "ovis"))
    with pytest.raises(ValueError):
        model(video("detection", "my_dataset"))
    model.video_unified_inference_enable = False
    model(video("sot", "sot_ytbvos18_val"))
    assert seen == ["entity", "vos"]
