#!/bin/bash
# round 6: the whole GPU suite and the bench line with generation 6 as the module path's MSDeformAttn core
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_i
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r06_i/bench.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "ms_per_step", "mask_logit_max_abs_err", "mask_sign_flips") if k in r})
print(json.dumps(r.get("roofline"), indent=1)[:1500])
print(r.get("roofline_msda_plus_mask_decode"))
print(r.get("steady_state_with_prompts"))
PY
UNIVS_MSDA_HEADS=0 python bench.py --steps 20 --warmup 5 > $O/bench_gen5.json 2> $O/bench_gen5.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r06_i/bench_gen5.json").read().strip().splitlines()[-1])
print("gen5:", {k: r[k] for k in ("value", "ms_per_step") if k in r}, r.get("roofline", {}).get("avg_launch_us"))
PY
