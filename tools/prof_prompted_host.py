"""Host side of the steady-state prompted clip (bench.py: steady_state_with_prompts): enqueue time per clip without synchronisation and a
cProfile of 10 clips, by own time and by cumulative time.      python tools/prof_prompted_host.py [--first]"""
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

first = "--first" in sys.argv
dev = torch.device("cuda:0")
swin, head = cases.build_model(dev)
case = dict(cases.CFG2, H=736, W=1280)
T = case["T"]
x = cases.preprocess(cases.cfg2_frames()).to(dev)
if first:
    tg0 = cases.targets_first_clip(cases.CFG2)[0]
else:
    tg0 = cases.targets_with_entities(case, first_frame_idx=1, n_ent=10)[0]
tvd = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in tg0.items()}


def clip():
    torch.manual_seed(0)
    tg = [dict(tvd)]
    if not first:
        head.prefetch_prompts(tg, T)
    return head(swin(x), targets=tg)


with torch.no_grad():
    for _ in range(3):
        clip()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        clip()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{'first' if first else 'prompted'} clip: host enqueue {(t1 - t0) / n * 1e3:.2f} ms per clip, with the final synchronisation {(t2 - t0) / n * 1e3:.2f} ms per clip")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        clip()
    pr.disable()
    torch.cuda.synchronize()
for key, rows in (("tottime", 45), ("cumulative", 70)):
    st = io.StringIO()
    pstats.Stats(pr, stream=st).sort_stats(key).print_stats(rows)
    print(f"==== by {key} ({n} clips)")
    for line in st.getvalue().splitlines():
        if line.strip() and ("ncalls" in line or "/" in line or "{" in line):
            print(line.replace(ROOT + "/", "")[:200])
