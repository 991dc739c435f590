#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_loop
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python tools/prof_video_loop.py --syncs 2>&1 | grep -v amdgpu | cut -c1-220 > $O/syncs.txt; cat $O/syncs.txt
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config5 --no-config4 --no-frame-sharded > $O/bench_loop.json 2> $O/bench_loop.err
python - <<PY
import json
d = json.loads(open("$O/bench_loop.json").read().strip().splitlines()[-1])
sl = d["sliding_clip_loop"]
for k, v in sl.items():
    if isinstance(v, dict): print(k, {kk: vv for kk, vv in v.items() if kk in ("ms_per_video", "frames_per_s", "window", "sampler")})
PY
