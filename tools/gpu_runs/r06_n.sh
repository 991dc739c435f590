#!/bin/bash
# round 6: the config tests with the library-Linear counters printed, the result-format tests on device tensors, the sampler test,
# and the bench line with the roofline objects of configs 4 and 5
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_n
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_modules_gpu.py tests/test_results_gpu.py -m gpu -q -s -k "config2_full or config4_full or config5_swinl or results_gpu or device_sampler or test_results" > $O/pytest.log 2>&1
grep -E "library GEMM|flipped|passed|failed|Error" $O/pytest.log | cut -c1-600
python bench.py --no-sliding-loop --no-frame-sharded > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r06_n/bench.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "ms_per_step")})
for c in ("config4_swinb_refvos", "config5_swinl_1080p"):
    d = r.get(c, {})
    print(c, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if k in ("ms_per_clip", "frames_per_s", "error")})
    for k in ("roofline", "roofline_mask_decode"):
        if k in d:
            print("  ", k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in d[k].items() if kk in ("frac", "avg_launch_us", "launches_per_step", "achieved", "queries", "tokens_per_frame", "impl")})
PY
tail -3 $O/bench.err
