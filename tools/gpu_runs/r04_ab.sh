#!/bin/bash
# round 4: A/B of one environment switch on ONE box: op / module tests first, then bench with VAR=0 / default, alternating
# usage: r04_ab.sh VAR "<ops -k expr>" "<modules -k expr>"
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_ab
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
if [ -n "$2" ]; then
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "$2" > $O/ops.log 2>&1
echo "pytest rc $?" >> $O/ops.log
tail -4 $O/ops.log
fi
if [ -n "$3" ]; then
timeout 900 python -m pytest tests/test_modules_gpu.py -q -m gpu -p no:cacheprovider -k "$3" > $O/parity.log 2>&1
echo "pytest rc $?" >> $O/parity.log
tail -3 $O/parity.log
fi
for i in 1 2 3; do
for v in 0 1; do
env $1=$v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_${v}_$i.json").read().strip().splitlines()[-1]); print("$1=$v run $i", round(d["value"],1), round(d["ms_per_step"],3), round(d["host_enqueue_ms_per_step"],2), d.get("mask_logit_max_abs_err"))
except Exception as e: print("$1=$v", "FAILED", e)
PY
done
done
echo done
