"""GPU: where the fused prompt-sampler prefix differs from the ATen formulation (debugging aid for tests/test_prompt_sampler_gpu.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.test_prompt_sampler_gpu import scene, encoder, run, S
dev = torch.device("cuda:0")
for Fk, n, hi, wi in [(2, 6, 16, 24), (2, 10, 92, 160)]:
    masks, boxes, feats, pos = scene(Fk, n, hi, wi, dev, seed=Fk * 100 + n)
    enc = encoder("device")
    pre_a, _, _ = run(enc, masks, boxes, feats, pos, fused=False)
    pre_f, _, _ = run(enc, masks, boxes, feats, pos, fused=True)
    h, w = masks.shape[-2:]
    for k in pre_a:
        if not torch.equal(pre_a[k], pre_f[k]):
            d = (pre_a[k] != pre_f[k])
            print(Fk, n, hi, wi, k, "differs in", int(d.sum()), "of", d.numel())
            if k == "sel":
                for f in range(Fk):
                    for e in range(n):
                        de = d[f, e]
                        if de.any():
                            ys, xs = de.nonzero()[:3].T
                            m = masks[f, e]
                            print("  entity", f, e, "pixels", int(de.sum()), "aten count", int(pre_a[k][f, e].sum()), "fused count", int(pre_f[k][f, e].sum()),
                                  "max", float(m.max()), "box", boxes[f, e].tolist(), "first", [(int(y), int(x), float(m[y, x])) for y, x in zip(ys, xs)])
