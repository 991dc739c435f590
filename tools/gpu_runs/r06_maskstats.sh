#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_loop
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mask_stats_gpu.py -q -m gpu -x > $O/pytest_maskstats.log 2>&1; echo "mask_stats rc $?"; tail -12 $O/pytest_maskstats.log
timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_vos_gpu.py -q -m gpu -x -k "loop or long_video or config3 or vos" > $O/pytest_loop.log 2>&1; echo "loop tests rc $?"; tail -3 $O/pytest_loop.log
timeout 600 python tools/prof_video_loop.py --syncs 2>&1 | grep -v amdgpu | cut -c1-220 > $O/syncs2.txt; head -12 $O/syncs2.txt
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config5 --no-config4 --no-frame-sharded > $O/bench_loop.json 2> $O/bench_loop.err
python - <<PY
import json
d = json.loads(open("$O/bench_loop.json").read().strip().splitlines()[-1])
sl = d["sliding_clip_loop"]
for k, v in sl.items():
    if isinstance(v, dict): print(k, {kk: vv for kk, vv in v.items() if kk in ("ms_per_video", "frames_per_s", "window", "sampler")})
print("value", d["value"], "enqueue", d["host_enqueue_ms_per_step"], "steady", d["steady_state_with_prompts"]["ms_per_clip"])
PY
