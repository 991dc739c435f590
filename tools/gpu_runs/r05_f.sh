#!/bin/bash
# round 5: V^T row stride in LDS (window attention, fused cross-attention): tests, kbench --only win and the bench line under both builds
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_f
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "window or cross_attention or deferred" > $O/tests.log 2>&1
echo "pytest rc $?" >> $O/tests.log; tail -3 $O/tests.log
for v in vtpad8 default vtpad8 default; do
  if [ $v = default ]; then L=""; else L="$R/univs_amd/libunivs_hip_$v.so"; fi
  UNIVS_HIP_LIB=$L timeout 300 python tools/kbench.py --only win 2>/dev/null | grep "f16x3" | awk -v t=$v '{print t, $1, $2, $3}' | sed 's/,//g' >> $O/win_$v.txt
  UNIVS_HIP_LIB=$L timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-config5 --no-config4 --no-sliding-loop --no-frame-sharded > $O/bench_$v.json 2>> $O/bench.err
  python - <<PY
import json
d = json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"], 1), round(d["ms_per_step"], 3), d.get("mask_logit_max_abs_err"), {k: round(v["avg_launch_us"],1) for k, v in d.get("roofline_window_attn", {}).get("stages", {}).items()} if isinstance(d.get("roofline_window_attn"), dict) and "stages" in d["roofline_window_attn"] else "")
PY
done
cat $O/win_vtpad8.txt | head -8; echo; cat $O/win_default.txt | head -8
