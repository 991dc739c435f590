"""Time one config-2 clip with N visual-prompt entities in the pool (second and third clip of a video)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from univs_amd import workloads as cases  # noqa: E402
from univs_amd import runtime  # noqa: E402

dev = torch.device("cuda:0")
runtime.enable_tuned_gemms()
swin, head = cases.build_model(dev)
case = dict(cases.CFG2, H=736, W=1280)
x = cases.preprocess(cases.cfg2_frames()).to(dev)
n_ent = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def targets(first_frame_idx):
    tv = cases.targets_with_entities(case, first_frame_idx=first_frame_idx, n_ent=n_ent)[0]
    return [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in tv.items()}]


with torch.no_grad():
    feats = swin(x)
    t1 = targets(1)
    for name, tg in (("first clip (no prompts)", [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in cases.targets_first_clip(case)[0].items()}]),
                     ("second clip, %d entities" % n_ent, t1)):
        for _ in range(2):
            tgc = [dict(tg[0])]
            torch.manual_seed(0)
            head(feats, targets=tgc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            tgc = [dict(tg[0])]
            torch.manual_seed(0)
            out = head(feats, targets=tgc)
        torch.cuda.synchronize()
        print(f"{name}: head {1e3 * (time.perf_counter() - t0) / 5:.2f} ms, queries {out['pred_masks'].shape[1]}")
