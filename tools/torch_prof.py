"""Where the ATen glue of one config-2 step goes: torch.profiler, every ATen op with device time of its own, aggregated by
(op, input shapes, calling line inside univs_amd).  `python tools/torch_prof.py [min_us]`"""
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from univs_amd import workloads as cases  # noqa: E402

dev = torch.device("cuda:0")
swin, head = cases.build_model(dev)
case = cases.CFG2
x = cases.preprocess(cases.cfg2_frames()).to(dev)
tg = lambda: [{k: (v.to(dev) if isinstance(v, torch.Tensor) and k != "frame_indices" else v) for k, v in cases.targets_first_clip(case)[0].items()}]  # noqa: E731
with torch.no_grad():
    for _ in range(2):
        head(swin(x), targets=tg())
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        head(swin(x), targets=tg())
        torch.cuda.synchronize()
min_us = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
agg = defaultdict(lambda: [0.0, 0])
for e in prof.events():
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = getattr(e, "self_cuda_time_total", 0)
    if not t or not e.name.startswith("aten::"):
        continue
    where = "-"
    for fr in (e.stack or []):
        if "univs_amd" in fr:
            where = fr.split("univs_amd/")[-1][:70]
            break
    shapes = str(getattr(e, "input_shapes", ""))[:90]
    k = (e.name, shapes, where)
    agg[k][0] += t
    agg[k][1] += 1
rows = sorted(((v[0], v[1], k) for k, v in agg.items()), reverse=True)
tot = sum(r[0] for r in rows)
print(f"# ATen ops with device time: {tot / 1e3:.3f} ms in {sum(r[1] for r in rows)} calls")
for t, c, (name, shapes, where) in rows:
    if t < min_us:
        break
    print(f"{t:9.1f} us  x{c:<3d} {name:28s} {where:72s} {shapes}")
