import os, sys, torch
sys.path.insert(0, "/root/repo")
from univs_amd import ops, synth
from univs_amd.switches import override
from tools.kbench import timeit
dev = torch.device("cuda:0")
for name, M, K, N, act, res in [("merge1", 73600, 384, 192, None, False), ("s3_qkv", 18400, 384, 1152, None, False), ("s3_fc1", 18400, 384, 1536, "gelu", False),
                                ("s3_proj", 18400, 384, 384, None, True), ("dec_kv_l8", 73600, 256, 768, None, False)]:
    x = synth.normal(f"gs/x/{M}x{K}", (M, K)).to(dev)
    w = synth.normal(f"gs/w/{N}x{K}", (N, K), std=K ** -0.5).to(dev)
    b = synth.normal(f"gs/b/{N}", (N,)).to(dev)
    r = synth.normal(f"gs/r/{M}x{N}", (M, N)).to(dev) if res else None
    fn = lambda: ops.linear_fused(x, w, b, act=act, residual=r)
    y0 = fn(); t0 = timeit(fn, iters=20, warmup=3) * 1e6
    with override(presplit_kmin=256):
        y1 = fn(); t1 = timeit(fn, iters=20, warmup=3) * 1e6
        with ops.configured(linear_ablate=6):
            t2 = timeit(fn, iters=20, warmup=3) * 1e6
    print(name, M, K, N, f"default {t0:.1f} us | presplit entry (tile) {t1:.1f} us, equal {torch.equal(y0, y1)} | presplit entry (pass kernel) {t2:.1f}", flush=True)
