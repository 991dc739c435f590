"""Which state of the g11a clip loop differs between two runs in device sampler mode, and by how much (tools/gpu_runs)."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("UNIVS_SAMPLER", "reference")
import torch
import cases, helpers
from tests.test_clip_loop_cpu import run_loop
from univs_amd.switches import SWITCHES, override

cuda = torch.device("cuda:0")
case = cases.LOOP_CASE
model = types.SimpleNamespace(backbone=helpers.build_swin(cuda), sem_seg_head=helpers.build_head(case, cuda))
enc = model.sem_seg_head.predictor.visual_prompt_sampler.visual_prompt_encoder

def runs(mode, n=3, **sw):
    enc.sampler_rng = mode
    outs = []
    with override(**sw):
        for _ in range(n):
            if hasattr(enc, "_dev_gen"):
                enc._dev_gen.clear()
            outs.append(run_loop(case, model, device=cuda, stability_score_thresh=0.0)[0])
    enc.sampler_rng = "reference"
    first = outs[0]
    rep = []
    for i, o in enumerate(outs[1:], 1):
        bad = [(k, (first[k].float() - o[k].float()).abs().max().item() if first[k].shape == o[k].shape else "shape") for k in first if not (first[k].shape == o[k].shape and torch.equal(first[k], o[k]))]
        rep.append(bad[:6])
    return rep

print("reference mode:", runs("reference"))
print("device mode:", runs("device"))
for name in sys.argv[1:]:
    print("device mode with", name, "off:", runs("device", **{name: False}))
# the head alone on clip 0's features, five times
x = cases.preprocess(cases.loop_frames(case)).to(cuda)
with torch.no_grad():
    feats = model.backbone(x[:3])
    outs = []
    for _ in range(5):
        t = cases.loop_targets(case)
        t[0].update(first_frame_idx=0, frame_indices=torch.arange(3, device=cuda)) if False else None
        o = model.backbone(x[:3])
        outs.append({k: v.clone() for k, v in o.items()})
print("backbone x5 identical:", all(torch.equal(outs[0][k], o[k]) for o in outs[1:] for k in o))
