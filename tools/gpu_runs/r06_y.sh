#!/bin/bash
# round 6: the fused MLP's weight stream by LDS-DMA against the register-staged form: parity tests, the GEMM launches of a clip under both builds, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_y
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "mlp" > $O/pytest_mlp.log 2>&1; tail -3 $O/pytest_mlp.log
python tools/gemmset.py --tag glds > $O/gemmset_glds.txt 2>&1
UNIVS_HIP_LIB=$R/univs_amd/libunivs_hip_mlp_regstage.so python tools/gemmset.py --tag regstage > $O/gemmset_regstage.txt 2>&1
grep -i "mlp\|ffn\|TOTAL" $O/gemmset_glds.txt | cut -c1-150
echo "--- register staging"
grep -i "mlp\|ffn\|TOTAL" $O/gemmset_regstage.txt | cut -c1-150
python bench.py --no-cpu-baseline --no-config5 --no-frame-sharded --no-config4 --no-sliding-loop > $O/bench.json 2> $O/bench.err
python - <<PY
import json
r = json.loads(open("gpurun_out/r06_y/bench.json").read().strip().splitlines()[-1])
print({k: r.get(k) for k in ("value", "ms_per_step", "mask_logit_max_abs_err", "mask_sign_flips", "host_enqueue_ms_per_step")})
print("  steady", r.get("steady_state_with_prompts"))
PY
