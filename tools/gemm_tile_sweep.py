"""The wide-K Linears of a clip on the two-dimensional tiled kernel (gemm_f16x3_tile.hip), tile shape and load depth forced through the
benchmark knobs of UnivsConfig (linear_grid_x = CT, linear_rows_per_pass = 64 RB, linear_ablate = 5 + NSLOT), against the pass kernel
(linear_ablate = 6).   python tools/gemm_tile_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univs_amd import ops, synth  # noqa: E402
from tools.kbench import timeit  # noqa: E402

SHAPES = [("s3_fc2_res", 18400, 1536, 384, True), ("s3_proj_res", 18400, 384, 384, True), ("s4_qkv", 4600, 768, 2304, False),
          ("s4_fc1", 4600, 768, 3072, False), ("s4_fc2_res", 4600, 3072, 768, True), ("s4_proj_res", 4600, 768, 768, True),
          ("merge2", 18400, 768, 384, False), ("merge3", 4600, 1536, 768, False)]
dev = torch.device("cuda:0")
QUICK = "--quick" in sys.argv        # (ablation builds: UNIVS_HIP_LIB=univs_amd/libunivs_hip_<nomfma|nosplit|...>.so)
if QUICK:
    SHAPES = [s for s in SHAPES if s[0] in ("s3_fc2_res", "s4_qkv", "merge2")]
for name, M, K, N, res in SHAPES:
    x = synth.normal(f"gs/x/{M}x{K}", (M, K)).to(dev)
    w = synth.normal(f"gs/w/{N}x{K}", (N, K), std=K ** -0.5).to(dev)
    b = synth.normal(f"gs/b/{N}", (N,)).to(dev)
    r = synth.normal(f"gs/r/{M}x{N}", (M, N)).to(dev) if res else None
    fn = lambda: ops.linear_fused(x, w, b, residual=r)
    out = []
    with ops.configured(linear_ablate=6):
        out.append(("pass", timeit(fn, iters=20, warmup=3) * 1e6))
    out.append(("auto", timeit(fn, iters=20, warmup=3) * 1e6))
    for ns in (2, 3, 4):
        if K % (32 * ns):
            continue
        for ct in (3, 4, 5):
            for rb in (2, 3, 4):
                if QUICK and (ns, ct, rb) not in ((4, 5, 3), (2, 4, 2), (4, 4, 3)):
                    continue
                with ops.configured(linear_ablate=5 + ns, linear_grid_x=ct, linear_rows_per_pass=64 * rb):
                    out.append((f"n{ns}c{ct}r{rb}", timeit(fn, iters=10, warmup=2) * 1e6))
    best = min(out[2:], key=lambda t: t[1])
    print(os.path.basename(os.environ.get("UNIVS_HIP_LIB", "default")), name, M, K, N, " ".join(f"{k}={v:.0f}" for k, v in out[:2]), "best", best[0], f"{best[1]:.0f}", "|", " ".join(f"{k}={v:.0f}" for k, v in out[2:]), flush=True)
