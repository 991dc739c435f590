"""Host time of one config-2 step (backbone + head, first clip) under cProfile: where the Python side of ~570 launches goes.
Sorted by own time (tottime): the wrappers, ATen calls and ctypes calls themselves."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from univs_amd import workloads as cases  # noqa: E402

dev = torch.device("cuda:0")
swin, head = cases.build_model(dev)
case = cases.CFG2
x = cases.preprocess(cases.cfg2_frames()).to(dev)
tg = lambda: [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in cases.targets_first_clip(case)[0].items()}]  # noqa: E731
with torch.no_grad():
    for _ in range(3):
        head(swin(x), targets=tg())
    torch.cuda.synchronize()
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        head(swin(x), targets=tg())
        t.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    print(f"host enqueue per step (no profiler): {sorted(t)[2] * 1e3:.2f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        head(swin(x), targets=tg())
        torch.cuda.synchronize()
    pr.disable()
for key in ("tottime", "cumulative"):
    st = io.StringIO()
    pstats.Stats(pr, stream=st).sort_stats(key).print_stats(40)
    print(f"==== by {key} (3 steps)")
    for line in st.getvalue().splitlines():
        if line.strip() and ("{" in line or "univs_amd" in line or "ncalls" in line or "torch" in line):
            print(line[:190])
