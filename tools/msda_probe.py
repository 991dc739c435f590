"""One variant of the head-major MSDA kernels at the config-2 geometry, launched `--iters` times (a rocprofv3 --pmc run sees one
kernel configuration per process):   python tools/msda_probe.py --gen 6 --cfg msda_sched=1,msda_strip_w=16,msda_strip_h=6"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from univs_amd import ops, synth
import cases

ap = argparse.ArgumentParser()
ap.add_argument("--gen", type=int, default=6)
ap.add_argument("--cfg", default="")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--T", type=int, default=5)
ap.add_argument("--geom", default="cfg2", choices=["cfg2", "cfg5", "cfg1"])
ap.add_argument("--std", type=float, default=2.0, help="std of the sampling offsets in pixels of the target level")
args = ap.parse_args()
dev = torch.device("cuda:0")
shapes = {"cfg2": [(23, 40), (46, 80), (92, 160)], "cfg5": [(34, 60), (68, 120), (136, 240)], "cfg1": [(8, 14), (16, 28), (32, 56)]}[args.geom]
T, M, L, P = args.T, 8, 3, 4
case = dict(name="kb", shapes=shapes, N=T, M=M, D=32, P=P, encoder=True, far=False)
value, shapes, lsi, loc, attn = cases.msda_inputs(case)
S = value.shape[1]
refs = []
for (h, w) in shapes:
    ys = (torch.arange(h, dtype=torch.float32) + 0.5) / h
    xs = (torch.arange(w, dtype=torch.float32) + 0.5) / w
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
    refs.append(torch.stack([xx.reshape(-1), yy.reshape(-1)], -1))
refp = torch.cat(refs, 0).view(1, S, 1, 2).expand(1, S, L, 2).contiguous()
norm = torch.tensor([[w, h] for (h, w) in shapes], dtype=torch.float32).view(1, 1, 1, L, 1, 2)
off = (loc - refp.view(1, S, 1, L, 1, 2)) * norm * (args.std / 2.0)
proj = torch.cat([off.reshape(T, S, -1), attn.clamp_min(1e-30).log().reshape(T, S, -1)], -1).contiguous().to(dev)
value = value.to(dev)
n_off = M * L * P * 2
refq = refp[:, :, 0].contiguous().to(dev)
cfg = {k: int(v) for k, v in (kv.split("=") for kv in args.cfg.split(",") if kv)}
if args.gen == 6:
    vhm, qhm = ops.msda_pack_heads(value, proj, n_off, shapes, P)
    fn = lambda: ops.msda_forward_heads(vhm, qhm, refq, shapes, lsi, M, P)
else:
    vhm, qhm = ops.msda_pack_head_major(value, proj, n_off, shapes, P)
    fn = lambda: ops.msda_forward_strips(vhm, qhm, refq, shapes, lsi, M, P)
with ops.configured(**cfg):
    out = fn()
    assert out is not None, "geometry not covered"
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(args.iters):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
alg = 3200.0 * S * T
print(json.dumps(dict(gen=args.gen, cfg=cfg, geom=args.geom, T=T, std=args.std, us_median=ts[len(ts) // 2], us_min=ts[0],
                      frac_hbm=alg / (ts[len(ts) // 2] * 1e-6) / 8e12, algorithmic_MB=alg / 1e6)))
