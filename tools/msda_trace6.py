"""Phase timeline of msda_fwd_heads from the instrumented build (python -m univs_amd.build --ablate heads_trace):
   UNIVS_HIP_LIB=univs_amd/libunivs_hip_heads_trace.so python tools/msda_trace6.py [--cfg msda_sched=1]
Per item and wave ten s_memtime stamps: 0 top, 1 loads issued, 2 records done, 3 gather stream done, 4 rare path + next inputs done,
5 every load arrived, 6 barrier A passed, 7 row commits issued, 8 reduction + output stores issued, 9 barrier B passed
(waves 0-3 and 4-7 listed apart: one of each per SIMD)."""
import argparse, ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from univs_amd import ops, _lib
import cases

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="")
args = ap.parse_args()
dev = torch.device("cuda:0")
shapes = [(23, 40), (46, 80), (92, 160)]
T, M, L, P = 5, 8, 3, 4
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
off = (loc - refp.view(1, S, 1, L, 1, 2)) * norm
proj = torch.cat([off.reshape(T, S, -1), attn.clamp_min(1e-30).log().reshape(T, S, -1)], -1).contiguous().to(dev)
value = value.to(dev)
refq = refp[:, :, 0].contiguous().to(dev)
cfg = {k: int(v) for k, v in (kv.split("=") for kv in args.cfg.split(",") if kv)}
vhm, qhm = ops.msda_pack_heads(value, proj, M * L * P * 2, shapes, P)
with ops.configured(**cfg):
    for _ in range(3):
        ops.msda_forward_heads(vhm, qhm, refq, shapes, lsi, M, P)
    torch.cuda.synchronize()
NI = 48
buf = np.zeros((256, NI, 8, 10), dtype=np.uint64)
lib = _lib.load()
rc = lib.univs_dbg_s6_trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_ulonglong(buf.nbytes))
assert rc == 0, rc
st = buf.astype(np.int64)
valid = st[:, :, 0, 0] > 0
names = ["issue input loads", "records", "gather stream (+ row requests)", "rare + next inputs", "wait loads", "barrier A", "commit rows (issue)", "reduce + stores", "barrier B"]
# s_memtime ticks at 100 MHz on gfx950?  report raw ticks and the share of an item
items = st[valid]                      # [n, 8 waves, 10]
d = np.diff(items, axis=2)             # [n, 8, 9]
total = items[:, :, 9] - items[:, :, 0]
print(f"items traced {items.shape[0]}, ticks per item (wave mean) {total.mean():.1f}  (min {total.min()}, max {total.max()})")
for grp, sl in (("waves 0-3 (X)", slice(0, 4)), ("waves 4-7 (Y)", slice(4, 8))):
    print(grp)
    for k, nm in enumerate(names):
        x = d[:, sl, k]
        print(f"  {nm:36s} mean {x.mean():9.1f}  p50 {np.median(x):9.1f}  p90 {np.percentile(x, 90):9.1f}   share {x.mean() / total.mean():6.3f}")
# gaps between consecutive items of a workgroup (segment changes: cold starts)
per_wg = []
for b in range(256):
    n = int(valid[b].sum())
    if n < 2: continue
    t = st[b, :n, 0, :]
    gaps = t[1:, 0] - t[:-1, 9]
    per_wg.append((t[-1, 9] - t[0, 0], gaps.sum(), gaps.max(), n))
per_wg = np.array(per_wg)
print(f"workgroups {len(per_wg)}: span of traced items mean {per_wg[:, 0].mean():.0f} ticks (min {per_wg[:, 0].min()}, max {per_wg[:, 0].max()}), "
      f"between-item gaps (cold starts) mean {per_wg[:, 1].mean():.0f} max single {per_wg[:, 2].max()}, items per workgroup mean {per_wg[:, 3].mean():.1f}")
# who is slow: the ten longest spans with their item / segment counts, and the XCD means (workgroup b runs on XCD b % 8)
rows = []
for b in range(256):
    n = int(valid[b].sum())
    if n < 1: continue
    t = st[b, :n, 0, :]
    gaps = t[1:, 0] - t[:-1, 9]
    nseg = 1 + int((gaps > 3000).sum())
    rows.append((int(t[-1, 9] - t[0, 0]), b, n, nseg, int(gaps[gaps > 3000].sum()) if n > 1 else 0, float((t[:, 9] - t[:, 0]).mean())))
rows.sort(reverse=True)
print("slowest workgroups (span, block, items, segments, cold gaps, mean item):", rows[:8])
print("fastest workgroups:", rows[-5:])
import collections
byx = collections.defaultdict(list)
for r in rows: byx[r[1] % 8].append(r)
for x in sorted(byx):
    v = byx[x]
    print(f"  XCD {x}: span mean {np.mean([r[0] for r in v]):.0f} max {max(r[0] for r in v)}, items mean {np.mean([r[2] for r in v]):.2f}, segments mean {np.mean([r[3] for r in v]):.2f}, item mean {np.mean([r[5] for r in v]):.0f}")
by = collections.defaultdict(list)
for r in rows: by[(r[2], r[3])].append(r[0])
print("span by (items, segments):", {k: (len(v), int(np.mean(v))) for k, v in sorted(by.items())})
t0 = st[valid][:, 0, 0].min()
ends = np.array([st[b, int(valid[b].sum()) - 1, 0, 9] for b in range(256) if valid[b].sum() > 0]) - t0
starts = np.array([st[b, 0, 0, 0] for b in range(256) if valid[b].sum() > 0]) - t0
print(f"first item starts: min {starts.min()} max {starts.max()}; last item ends: min {ends.min()} mean {ends.mean():.0f} max {ends.max()}")
