"""Run-to-run bit identity of the head-major MSDA kernels on small and large pyramids (tools/gpu_runs): every geometry 20 times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from univs_amd import ops, synth
import cases

dev = torch.device("cuda:0")
GEOMS = {"loop": [(2, 3), (4, 6), (8, 12)], "cfg1": [(8, 14), (16, 28), (32, 56)], "ragged": [(5, 7), (9, 13), (17, 25)],
         "cfg2": [(23, 40), (46, 80), (92, 160)]}
bad = 0
for name, shapes in GEOMS.items():
    for T in (3, 5):
        case = dict(name="det" + name, shapes=shapes, N=T, M=8, D=32, P=4, encoder=True, far=True)
        value, shapes_, lsi, loc, attn = cases.msda_inputs(case)
        S = value.shape[1]
        L, M, P = len(shapes), 8, 4
        refs = []
        for (h, w) in shapes:
            ys = (torch.arange(h, dtype=torch.float32) + 0.5) / h
            xs = (torch.arange(w, dtype=torch.float32) + 0.5) / w
            yy, xx = torch.meshgrid(ys, xs, indexing="ij")
            refs.append(torch.stack([xx.reshape(-1), yy.reshape(-1)], -1))
        refp = torch.cat(refs, 0).view(1, S, 1, 2).expand(1, S, L, 2).contiguous()
        norm = torch.tensor([[w, h] for (h, w) in shapes], dtype=torch.float32).view(1, 1, 1, L, 1, 2)
        off = (loc - refp.view(1, S, 1, L, 1, 2)) * norm
        proj = torch.cat([off.reshape(T, S, -1), attn.clamp_min(1e-30).log().reshape(T, S, -1)], -1).contiguous().to(dev)
        value = value.to(dev)
        refq = refp[:, :, 0].contiguous().to(dev)
        for gen in (6, 5):
            if gen == 6:
                vhm, qhm = ops.msda_pack_heads(value, proj, M * L * P * 2, shapes, P)
                fn = lambda: ops.msda_forward_heads(vhm, qhm, refq, shapes, lsi, M, P)
            else:
                vhm, qhm = ops.msda_pack_head_major(value, proj, M * L * P * 2, shapes, P)
                fn = lambda: ops.msda_forward_strips(vhm, qhm, refq, shapes, lsi, M, P)
            first = fn()
            if first is None:
                print(name, T, gen, "not covered"); continue
            l_, a_ = ops.msda_prepare(proj, M * L * P * 2, refp.to(dev), shapes, M, L, P)
            with ops.configured(msda_impl=1):
                want = ops.ms_deform_attn_forward(value, shapes, lsi, l_, a_)
            ndiff, mx = 0, 0.0
            for _ in range(20):
                again = fn()
                if not torch.equal(first, again):
                    ndiff += 1
                    mx = max(mx, (first - again).abs().max().item())
            print(f"{name:7s} T={T} gen {gen}: runs differing from the first {ndiff}/20 (max abs {mx:.2e}); vs generic {(first - want).abs().max().item():.2e}")
            bad += ndiff
print("BAD" if bad else "all identical")
