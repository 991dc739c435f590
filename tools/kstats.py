"""Summarise a rocprofv3 --kernel-trace --stats CSV: short kernel names, calls, total ms, avg us, %."""
import csv
import glob
import re
import sys

root = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
files = glob.glob(root + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    if n.startswith("Cijk_"):
        m = re.search(r"MT(\d+x\d+x\d+)", n)
        return "hipBLASLt GEMM " + n[:22] + (" MT" + m.group(1) if m else "")
    n = re.sub(r"at::native::", "", n)
    return n[:90]


tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
agg = {}
for r in rows:
    k = short(r["Name"])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += int(r["Calls"])
    a[1] += float(r["TotalDurationNs"])
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t / tot * 100:6.2f}%  {t / 1e6:9.3f} ms  {c:6d} calls  {t / c / 1e3:9.1f} us  {k}")
