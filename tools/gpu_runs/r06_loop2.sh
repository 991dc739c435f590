#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_loop
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_modules_gpu.py -q -m gpu -x -k "loop or long_video or config3" > $O/pytest_loop.log 2>&1; echo "loop tests rc $?"; tail -3 $O/pytest_loop.log
timeout 600 python tools/prof_video_loop.py 2>&1 | grep -v amdgpu | cut -c1-220 > $O/stages_after.txt; cat $O/stages_after.txt
timeout 600 python tools/prof_video_loop.py --cprofile-post detect_newly_entities_per_clip_instance 2>&1 | grep -v amdgpu | cut -c1-200 | sed -n 12,30p
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config5 --no-config4 --no-frame-sharded > $O/bench_loop.json 2> $O/bench_loop.err
python - <<PY
import json
d = json.loads(open("$O/bench_loop.json").read().strip().splitlines()[-1])
print(json.dumps(d.get("sliding_clip_loop"))[:1500])
PY
