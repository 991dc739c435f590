#!/bin/bash
# round 6: 12 waves per workgroup in the 7 x 7 window-attention kernel (3 per SIMD: 156 VGPRs, 122 KB of LDS) against 8
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_u
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_modules_gpu.py -m gpu -q -k "window or winattn or swin_matches or config2_full" 2>&1 | tail -3
python bench.py --no-config5 --no-config4 --no-sliding-loop --no-frame-sharded --no-cpu-baseline > $O/bench.json 2>/dev/null
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r06_u/bench.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "ms_per_step")})
w = r.get("roofline_window_attn", {})
print({k: (round(v["avg_launch_us"], 1), round(v["frac"], 3)) for k, v in w.get("per_stage", {}).items()}, w.get("ms_per_clip"))
PY
