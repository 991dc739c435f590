"""round 4, run Z: linear_f16x3 with 144 features per pass (RB = 9) against 128: the encoder's 288-column projection, Swin stage-2 qkv"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from univs_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


res = {}
for nm, M, K, N in (("enc_query_proj", 96600, 256, 288), ("swin_s2_qkv", 73600, 192, 576)):
    x = synth.normal(f"z/x/{M}x{K}", (M, K)).to(dev)
    w = synth.normal(f"z/w/{N}x{K}", (N, K), std=K ** -0.5).to(dev)
    b = synth.normal(f"z/b/{N}", (N,)).to(dev)
    row = {}
    ref = None
    for rpp in (128, 0, 128, 0):
        with ops.configured(linear_rows_per_pass=rpp):
            y = ops.linear_fused(x, w, b)
            if ref is None:
                ref = y
            row.setdefault(f"rows_per_pass_{rpp or 144}", []).append(round(timeit(lambda: ops.linear_fused(x, w, b)) * 1e6, 1))
            row[f"max_abs_diff_{rpp or 144}"] = float((y - ref).abs().max())
    if nm == "enc_query_proj":
        for rpp in (128, 0, 128, 0):
            with ops.configured(linear_rows_per_pass=rpp):
                row.setdefault(f"blocked_{rpp or 144}", []).append(round(timeit(lambda: ops.linear_blocked(x.view(5, 19320, K), w, b, 19320, 36)) * 1e6, 1))
    res[nm] = row
print(json.dumps(res, indent=1))
