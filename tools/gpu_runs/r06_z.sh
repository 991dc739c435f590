#!/bin/bash
# round 6: the phase-shifted fused MLP (waves 4-7 half a chunk behind waves 0-3) against the lockstep kernel: parity tests, the MLP launches of a clip, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_z
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "mlp" > $O/pytest_mlp.log 2>&1; tail -5 $O/pytest_mlp.log
python tools/gemmset.py --tag phase_shifted > $O/gemmset_ps.txt 2>&1
python tools/gemmset.py --tag lockstep --linear-ablate 8 > $O/gemmset_lockstep.txt 2>&1
grep -i "mlp\|ffn\|TOTAL" $O/gemmset_ps.txt | cut -c1-150
grep -i "mlp\|ffn\|TOTAL" $O/gemmset_lockstep.txt | cut -c1-150
python bench.py --no-cpu-baseline --no-config5 --no-frame-sharded --no-config4 --no-sliding-loop > $O/bench.json 2> $O/bench.err
python - <<PY
import json
r = json.loads(open("gpurun_out/r06_z/bench.json").read().strip().splitlines()[-1])
print({k: r.get(k) for k in ("value", "ms_per_step", "mask_logit_max_abs_err", "mask_sign_flips", "host_enqueue_ms_per_step")})
PY
