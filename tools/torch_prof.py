"""Where the ATen glue of one config-2 step goes: torch.profiler, aggregated by (op, input shapes)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from univs_amd import workloads as cases  # noqa: E402

dev = torch.device("cuda:0")
swin, head = cases.build_model(dev)
case = cases.CFG2
x = cases.preprocess(cases.cfg2_frames()).to(dev)
tg = lambda: [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in cases.targets_first_clip(case)[0].items()}]
with torch.no_grad():
    for _ in range(2):
        head(swin(x), targets=tg())
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
        head(swin(x), targets=tg())
        torch.cuda.synchronize()
want = sys.argv[1:] or ["aten::add", "aten::copy_", "aten::add_", "aten::div", "aten::mul", "aten::roll", "aten::masked_fill", "aten::gelu",
                         "aten::native_group_norm", "aten::_softmax", "aten::cat", "aten::clone"]
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in want:
        t = getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0)
        rows.append((t, e.key, e.count, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
for t, k, c, s in rows[:45]:
    print(f"{t / 1e3:8.3f} ms  {k:24s} x{c:<4d} {s}")
