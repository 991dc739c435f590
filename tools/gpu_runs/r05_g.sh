#!/bin/bash
# round 5: the few-rows MLP chain (mask-embedding MLP + decoder_norm in one launch): op test, head / model goldens, A/B of the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_g
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "small_mlp or small_linear" > $O/ops.log 2>&1
echo "pytest rc $?" >> $O/ops.log; tail -4 $O/ops.log
timeout 900 python -m pytest tests/test_modules_gpu.py -q -m gpu -p no:cacheprovider -k "head or config2 or g4_g5 or clip_loop or device_sampler" > $O/mod.log 2>&1
echo "pytest rc $?" >> $O/mod.log; tail -4 $O/mod.log
for i in 1 2; do
for v in "UNIVS_SMALL_MLP_CHAIN=0" "UNIVS_SMALL_MLP_NORM=0" "UNIVS_SMALL_MLP_CHAIN=1"; do
  env $v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-config5 --no-config4 --no-sliding-loop --no-frame-sharded > $O/bench.json 2>> $O/bench.err
  python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("$v", round(d["value"], 1), round(d["ms_per_step"], 3), round(d["host_enqueue_ms_per_step"], 2), d.get("mask_logit_max_abs_err"), d.get("steady_state_with_prompts", {}).get("ms_per_clip"))
PY
done
done
