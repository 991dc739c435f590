#!/bin/bash
# round 4, GPU call R: PatchMerging gather + norm kernel: op test, Swin goldens, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_r
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "${1:-patch_merge}" > $O/ops.log 2>&1
echo "pytest rc $?" >> $O/ops.log
tail -4 $O/ops.log
timeout 900 python -m pytest tests/test_modules_gpu.py -q -m gpu -p no:cacheprovider -k "${2:-swin or config2 or config4}" > $O/parity.log 2>&1
echo "pytest rc $?" >> $O/parity.log
tail -3 $O/parity.log
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench$i.json 2> $O/bench$i.err
python - <<PY
import json
try:
    d=json.loads(open("$O/bench$i.json").read().strip().splitlines()[-1]); print("bench$i", round(d["value"],1), round(d["ms_per_step"],3), d["host_enqueue_ms_per_step"], d.get("mask_logit_max_abs_err"))
except Exception as e: print("bench$i", "FAILED", e)
PY
done
echo done
