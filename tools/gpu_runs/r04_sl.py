"""round 4: the few-rows Linear kernel against the library GEMM (+ the elementwise launches it absorbs), decoder shapes"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from univs_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
F = torch.nn.functional


def timeit(fn, iters=200, warmup=20):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / iters * 1e6, 2)


res = {}
M = 500
x = synth.normal("slb/x", (M, 256)).to(dev)
pos = synth.normal("slb/p", (M, 256)).to(dev)
r = synth.normal("slb/r", (M, 256)).to(dev)
g = torch.ones(256, device=dev)
b0 = torch.zeros(256, device=dev)
for nm, K, N in (("256x256", 256, 256), ("256x768", 256, 768), ("256x2048", 256, 2048), ("2048x256", 2048, 256)):
    xx = x if K == 256 else synth.normal("slb/x2048", (M, 2048)).to(dev)
    w = synth.normal(f"slb/w{K}x{N}", (N, K), std=K ** -0.5).to(dev)
    b = synth.normal(f"slb/b{N}", (N,)).to(dev)
    row = {"small_linear": timeit(lambda: ops.small_linear(xx, w, b)), "F.linear": timeit(lambda: F.linear(xx, w, b))}
    if N == 256:
        row["small_linear_res_ln"] = timeit(lambda: ops.small_linear(xx, w, b, residual=r, ln=(g, b0, 1e-5)))
        row["F.linear_then_ln"] = timeit(lambda: ops.layer_norm(F.linear(xx, w, b), g, b0, 1e-5, residual=r))
    if K == 256:
        row["small_linear_add"] = timeit(lambda: ops.small_linear(xx, w, b, x_add=pos))
        row["add_then_F.linear"] = timeit(lambda: F.linear(xx + pos, w, b))
    res[nm] = row
print(json.dumps(res, indent=1))
