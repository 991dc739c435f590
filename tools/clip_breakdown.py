"""Steady-state per-clip kernel breakdown from a rocprofv3 kernel trace of bench.py.

rocprofv3's --stats summary covers the whole process, warm-up included (MIOpen's find pass alone launches seconds of
naive reference convolutions).  This takes the kernel-trace CSV, uses an anchor kernel launched a fixed number of times
per clip (the MSDeformAttn forward: 6 encoder layers) to cut the trace into clips, keeps the last `--last` clips and
prints the time per clip by kernel, plus the GPU-idle time between kernels.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline
    python tools/clip_breakdown.py gpurun_out/prof/*/*_kernel_trace.csv --skip 4 --last 8 > profiles/rNN_bench_clip_breakdown.txt
(--skip 4: the 3 warm-up clips and the first timed one; the clips after the timed loop belong to bench.py's instrumented
per-operator pass, which synchronises after every operator)
"""
import argparse
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--anchor", default="msda_fwd")
    ap.add_argument("--per-clip", type=int, default=6)
    ap.add_argument("--last", type=int, default=8, help="clips to average over")
    ap.add_argument("--skip", type=int, default=-1, help="clips to skip from the start (default: take the LAST clips of the trace)")
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--timeline", action="store_true", help="every launch of ONE clip in order: start (us from the clip's first "
                    "launch), duration, gap to the previous kernel's end, name")
    ap.add_argument("--detail", default="", help="substring: list every launch of matching kernels in ONE clip with its neighbours")
    args = ap.parse_args()
    rows = []
    with open(args.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    anchors = [i for i, r in enumerate(rows) if args.anchor in r[2]]
    nclips = len(anchors) // args.per_clip
    if nclips < args.last + 1:
        sys.exit(f"only {nclips} clips in the trace")
    # a clip = from the first anchor launch of clip c to the first anchor launch of clip c + 1 (same phase of every clip)
    c0 = nclips - args.last - 1 if args.skip < 0 else args.skip
    if c0 + args.last >= nclips:
        sys.exit(f"clips {c0}..{c0 + args.last} requested, {nclips} in the trace")
    first = anchors[c0 * args.per_clip]
    end = anchors[(c0 + args.last) * args.per_clip]
    seg = rows[first:end]
    if args.timeline:
        one = rows[first:anchors[(c0 + 1) * args.per_clip]]
        print(f"# clip {c0}: {len(one)} launches;   start_us   dur_us   gap_us  kernel")
        t0, prev_end = one[0][0], one[0][0]
        for s_, e_, n_ in one:
            print(f"  {(s_ - t0) / 1e3:10.1f} {(e_ - s_) / 1e3:8.1f} {(s_ - prev_end) / 1e3:8.1f}  {short(n_)[:90]}")
            prev_end = max(prev_end, e_)
        return
    if args.detail:
        one = rows[first:anchors[(c0 + 1) * args.per_clip]]
        print(f"# launches matching {args.detail!r} in clip {c0} (us, previous kernel -> next kernel)")
        for i, (s_, e_, n_) in enumerate(one):
            if args.detail in n_ and (e_ - s_) > 8000:
                prev = short(one[i - 1][2])[:60] if i else "-"
                nxt = short(one[i + 1][2])[:60] if i + 1 < len(one) else "-"
                print(f"  {(e_ - s_) / 1e3:8.1f}  {prev}  ->  {nxt}")
        return
    span = (seg[-1][1] - seg[0][0]) / args.last
    busy = defaultdict(float)
    calls = defaultdict(int)
    tot = 0.0
    for s, e, n in seg:
        busy[short(n)] += (e - s)
        calls[short(n)] += 1
        tot += e - s
    print(f"# {args.csv}: clips {c0}..{c0 + args.last - 1} of {nclips}; {len(seg) / args.last:.0f} launches per clip")
    print(f"# per clip: wall {span / 1e6:.3f} ms, kernels busy {tot / args.last / 1e6:.3f} ms, idle between kernels {(span - tot / args.last) / 1e6:.3f} ms")
    print(f"# {'ms/clip':>9} {'%busy':>6} {'calls/clip':>10} {'us/call':>9}  kernel")
    for n, t in sorted(busy.items(), key=lambda kv: -kv[1])[:args.top]:
        print(f"  {t / args.last / 1e6:9.3f} {100 * t / tot:6.2f} {calls[n] / args.last:10.1f} {t / calls[n] / 1e3:9.1f}  {n}")
    rest = sorted(busy.items(), key=lambda kv: -kv[1])[args.top:]
    print(f"  {sum(t for _, t in rest) / args.last / 1e6:9.3f} {100 * sum(t for _, t in rest) / tot:6.2f} {sum(calls[n] for n, _ in rest) / args.last:10.1f} {'':>9}  ({len(rest)} more kernels)")


if __name__ == "__main__":
    main()
