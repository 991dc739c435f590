#!/bin/bash
# round 6: (1) matrix-pipe / vector-ALU / LDS overlap probe; (2) the bench line with and without hipGraph replay of backbone + pixel decoder
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_v
mkdir -p $O
cd $R
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_overlap.hip -o /tmp/mfma_overlap 2>/dev/null && /tmp/mfma_overlap > $O/mfma_overlap.txt 2>&1
cat $O/mfma_overlap.txt
for g in 0 1; do
  UNIVS_GRAPHS=$g python bench.py --no-cpu-baseline --no-config5 --no-frame-sharded --no-config4 --no-sliding-loop > $O/bench_g$g.json 2> $O/bench_g$g.err
  python - <<PY
import json
r = json.loads(open("gpurun_out/r06_v/bench_g$g.json").read().strip().splitlines()[-1])
print("graphs=$g", {k: r.get(k) for k in ("value", "ms_per_step", "mask_logit_max_abs_err", "mask_sign_flips", "host_enqueue_ms_per_step")})
print("  steady", r.get("steady_state_with_prompts"))
PY
done
