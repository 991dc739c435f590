"""round 4: key segments of the fused attention kernel for short key sequences (the decoder's self-attention: 500 x 500)"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from univs_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=30, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


res = {}
for nm, L, S, N in (("self_attn_500", 500, 500, 1), ("self_attn_2000", 2000, 2000, 1), ("cross_920", 100, 920, 5), ("cross_3680", 100, 3680, 5),
                    ("cross_14720", 100, 14720, 5)):
    q = synth.normal(f"sg/q/{L}x{N}", (L, N, 256)).to(dev)
    k = synth.normal(f"sg/k/{S}x{N}", (S, N, 256)).to(dev)
    v = synth.normal(f"sg/v/{S}x{N}", (S, N, 256)).to(dev)
    m = (torch.rand(N, L, S, generator=torch.Generator().manual_seed(3)) < 0.5).to(dev)
    ref = ops.cross_attention(q, k, v, m, 8, 32 ** -0.5)
    row = {}
    for seg in (0, 2, 4, 8, 16, 32, 64):
        with ops.configured(xattn_segments=seg):
            y = ops.cross_attention(q, k, v, m, 8, 32 ** -0.5)
            row[f"seg{seg}"] = round(timeit(lambda: ops.cross_attention(q, k, v, m, 8, 32 ** -0.5)) * 1e6, 1)
            row[f"err{seg}"] = float((y - ref).abs().max())
    res[nm] = row
print(json.dumps(res, indent=1))
