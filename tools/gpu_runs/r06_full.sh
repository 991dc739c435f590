#!/bin/bash
# round 6: the whole GPU suite, smoke, and the driver's bench command
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_full
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r06_full/bench.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "ms_per_step", "mask_logit_max_abs_err", "mask_sign_flips", "host_enqueue_ms_per_step")})
print("roofline", {k: r["roofline"][k] for k in ("frac", "avg_launch_us", "traffic")})
print("steady", r.get("steady_state_with_prompts"))
print("sliding", json.dumps(r.get("sliding_clip_loop"))[:500])
print("cpu", r.get("cpu_baseline", {}).get("value"))
PY
