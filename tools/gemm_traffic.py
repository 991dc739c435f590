"""A handful of the clip's three-product GEMM launches, a few iterations each, for a rocprofv3 --pmc run (tools/gpu_runs/r05_j.sh):
how many of the x re-reads of the passes over N reach the fabric.  `--summarise DIR` prints mean counters per (kernel, grid)."""
import argparse
import csv
import glob
import os
import sys
from collections import defaultdict

SHAPES = [  # name, M, K, N, residual
    ("s3_fc2_res", 18400, 1536, 384, True),
    ("s3_proj_res", 18400, 384, 384, True),
    ("s3_qkv", 18400, 384, 1152, False),
    ("s1_proj_res", 294400, 96, 96, True),
    ("enc_output_proj_res", 96600, 256, 256, True),
    ("dec_kv_l8", 73600, 256, 768, False),
    ("s4_fc2_res", 4600, 3072, 768, True),
]


def run(args):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from univs_amd import ops, synth
    if args.linear_ablate:
        ops.configure(linear_ablate=args.linear_ablate)
    dev = torch.device("cuda:0")
    for name, M, K, N, res in SHAPES:
        x = synth.normal(f"gs/x/{M}x{K}", (M, K)).to(dev)
        w = synth.normal(f"gs/w/{N}x{K}", (N, K), std=K ** -0.5).to(dev)
        b = synth.normal(f"gs/b/{N}", (N,)).to(dev)
        r = synth.normal(f"gs/r/{M}x{N}", (M, N)).to(dev) if res else None
        for _ in range(args.iters):
            ops.linear_fused(x, w, b, residual=r)
        torch.cuda.synchronize()
        print(name, M, K, N, "algorithmic MB", round(4e-6 * M * (K + N + (N if res else 0)), 1), flush=True)


def summarise(root):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "f16x3" not in k or "presplit" in k:
                continue
            acc[(k.split("(")[0][-40:], row.get("Grid_Size", "?"))][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for (k, g), d in sorted(acc.items()):
        print(f"{k} grid {g}: " + "  ".join(f"{c}={sum(v) / len(v):.4g} (n={len(v)})" for c, v in sorted(d.items())))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--linear-ablate", type=int, default=0)
    ap.add_argument("--summarise", default=None)
    a = ap.parse_args()
    summarise(a.summarise) if a.summarise else run(a)
