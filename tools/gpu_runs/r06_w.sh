#!/bin/bash
# round 6: the prompted clip after the sampler kernels: module / loop tests, the ATen operator count by source line, the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_w
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_modules_gpu.py tests/test_prompt_sampler_gpu.py tests/test_vos_gpu.py -q -x > $O/pytest.log 2>&1
tail -5 $O/pytest.log
python tools/launch_sources.py > $O/launch_sources_prompted.txt 2> $O/err.txt
head -45 $O/launch_sources_prompted.txt | cut -c1-170
python bench.py --no-cpu-baseline --no-config5 --no-frame-sharded --no-config4 --no-sliding-loop > $O/bench.json 2> $O/bench.err
python - <<PY
import json
r = json.loads(open("gpurun_out/r06_w/bench.json").read().strip().splitlines()[-1])
print({k: r.get(k) for k in ("value", "ms_per_step", "mask_logit_max_abs_err", "mask_sign_flips", "host_enqueue_ms_per_step")})
print("  steady", r.get("steady_state_with_prompts"))
PY
