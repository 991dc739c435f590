#!/bin/bash
# the sliding clip loop under rocprofv3's kernel trace: device busy time of a video against its wall time (is the loop host-bound?)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_loop
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ltrace -o t -- python $R/tools/prof_video_loop.py > $O/looptrace_stdout.txt 2> $O/looptrace.err
CSV=$(find $O/ltrace -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$CSV")))
ts = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# the second video = the second half of the launches by time (the tool runs one warm-up video and one timed video after model set-up)
n = len(ts)
print("kernel launches traced:", n)
# find the two videos: the longest gaps do not help; take launches of the last video by counting patch_embed4 launches (one per window = per video)
pe = [i for i, t in enumerate(ts) if "patch_embed4" in t[2]]
print("backbone runs:", len(pe))
start = pe[-1]
seg = ts[start:]
busy = sum(e - s for s, e, _ in seg) / 1e6
wall = (seg[-1][1] - seg[0][0]) / 1e6
print(f"last video: {len(seg)} launches, device busy {busy:.1f} ms of {wall:.1f} ms between its first and last kernel ({100 * busy / wall:.0f} %)")
import collections
by = collections.Counter()
for s, e, k in seg:
    by[k.split("(")[0][:70]] += (e - s) / 1e6
for k, v in by.most_common(12):
    print(f"  {v:7.2f} ms  {k}")
PY
grep -v amdgpu $O/looptrace_stdout.txt | head -3
rm -rf $O/ltrace
