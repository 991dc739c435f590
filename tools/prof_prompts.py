import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from univs_amd import workloads as cases
dev = torch.device("cuda:0")
swin, head = cases.build_model(dev)
case = dict(cases.CFG2, H=736, W=1280)
x = cases.preprocess(cases.cfg2_frames()).to(dev)
tv0 = cases.targets_with_entities(case, first_frame_idx=1, n_ent=10)[0]
tvd = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in tv0.items()}
mk = lambda: [dict(tvd)]
from torch.profiler import ProfilerActivity, profile
with torch.no_grad():
    feats = swin(x)
    for _ in range(2):
        head(feats, targets=mk())
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
        t0 = time.perf_counter()
        head(feats, targets=mk())
        torch.cuda.synchronize()
        print("wall ms", 1e3 * (time.perf_counter() - t0))
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
    if t > 60:
        rows.append((t, e.key, e.count, str(e.input_shapes)[:100]))
rows.sort(reverse=True)
print('total device ms', sum(r[0] for r in rows if not r[1].startswith('aten::') and 'hip' not in r[1]) / 1e3)
for t, k, c, sh in rows[:70]:
    print(f"{t / 1e3:8.3f} ms  {k[:40]:40s} x{c:<4d} {sh}")
