"""The encoder FFN as the model calls it (norm1 on the x tile, its output as the residual, norm2 on the result, the next layer's `src + pos`:
msdeformattn.py:87-95) and in its plain form, timed alone.   [UNIVS_HIP_LIB=univs_amd/libunivs_hip_plainsplit.so] python tools/ffn_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univs_amd import ops, synth
from tools.kbench import timeit
dev = torch.device("cuda:0")
M, C, Hd = 5 * 19320, 256, 1024
x = synth.normal("fp/x", (M, C)).to(dev)
w1 = synth.normal("fp/w1", (Hd, C), std=C ** -0.5).to(dev); b1 = synth.normal("fp/b1", (Hd,), std=0.5).to(dev)
w2 = synth.normal("fp/w2", (C, Hd), std=Hd ** -0.5).to(dev); b2 = synth.normal("fp/b2", (C,), std=0.5).to(dev)
g1 = (1 + 0.1 * synth.normal("fp/g1", (C,))).to(dev); be1 = (0.1 * synth.normal("fp/be1", (C,))).to(dev)
g2 = (1 + 0.1 * synth.normal("fp/g2", (C,))).to(dev); be2 = (0.1 * synth.normal("fp/be2", (C,))).to(dev)
pos = synth.normal("fp/pos", (19320, C)).to(dev)
tag = os.path.basename(os.environ.get("UNIVS_HIP_LIB", "default"))
f_plain = lambda: ops.mlp_fused(x, w1, b1, w2, b2, "relu")
f_model = lambda: ops.mlp_fused(x, w1, b1, w2, b2, "relu", ln=(g1, be1, 1e-5), residual_normed=True, post_ln=(g2, be2, 1e-5), post_add=pos)
f_res = lambda: ops.mlp_fused(x, w1, b1, w2, b2, "relu", residual=x, post_ln=(g2, be2, 1e-5), post_add=pos)
for nm, f in (("plain", f_plain), ("residual + post-LN + pos", f_res), ("as in the model (LN in, LN residual, post-LN, pos)", f_model)):
    y = f()
    print(tag, nm, "None" if y is None else f"{timeit(f, iters=20, warmup=4) * 1e6:.1f} us", flush=True)
