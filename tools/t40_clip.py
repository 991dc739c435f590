"""The 40-frame 720p clip of bench.py's `frame_sharded_n1` timed apart (backbone / head) per window-attention mode."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univs_amd import workloads as cases, runtime, synth
dev = torch.device("cuda:0")
runtime.enable_tuned_gemms()
swin, head = cases.build_model(dev)
case = cases.CFG2
T40 = 40
fr = synth.synthetic_frames(T40, case["H"], case["W"], "frames/t40").to(dev)
mean = torch.tensor([123.675, 116.28, 103.53], device=dev).view(1, 3, 1, 1)
std = torch.tensor([58.395, 57.12, 57.375], device=dev).view(1, 3, 1, 1)
tv = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in cases.targets_first_clip(case)[0].items()}
tv["frame_indices"] = torch.arange(T40, device=dev)
@torch.no_grad()
def step():
    x = torch.nn.functional.pad((fr - mean) / std, (0, 0, 0, 16))
    f = swin(x)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    o = head(f, targets=[dict(tv)])
    torch.cuda.synchronize()
    return t1, o
for mma in ("f16x3", "f32", "f16x3"):
    swin.set_attention_mma(mma)
    step(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); t1, _ = step(); t2 = time.perf_counter()
        ts.append((round((t1 - t0) * 1e3, 1), round((t2 - t1) * 1e3, 1)))
    print(mma, "per clip (backbone ms, head ms):", ts, flush=True)
