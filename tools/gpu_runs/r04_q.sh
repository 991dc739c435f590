#!/bin/bash
# round 4, GPU call Q: proj + MLP with W0 in the MLP's weight stream (pair chunks); 32 rows per wave / one wave per SIMD variants
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_q
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -s -p no:cacheprovider -k "proj_mlp or mlp_fused" > $O/ops.log 2>&1
echo "pytest rc $?" >> $O/ops.log
grep -E "proj_mlp_fused|passed|failed|rc |Error|error" $O/ops.log | tail -14
timeout 600 python tools/kbench.py --only mlp > $O/kbench_mlp.txt 2> $O/kbench_mlp.err
grep -E "encoder_ffn|swin_s" $O/kbench_mlp.txt | cut -c1-900
UNIVS_FUSED_PROJ_MLP=1 timeout 900 python -m pytest tests/test_modules_gpu.py -q -m gpu -p no:cacheprovider -k "swin or pixel_decoder or g2 or config2" > $O/parity.log 2>&1
echo "pytest rc $?" >> $O/parity.log
tail -3 $O/parity.log
UNIVS_FUSED_PROJ_MLP=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench_on.json 2> $O/bench_on.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench_off.json 2> $O/bench_off.err
UNIVS_FUSED_PROJ_MLP=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench_on2.json 2> $O/bench_on2.err
for f in bench_on bench_off bench_on2; do python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1]); print("$f", round(d["value"],1), round(d["ms_per_step"],3), d["host_enqueue_ms_per_step"], d.get("mask_logit_max_abs_err"))
except Exception as e: print("$f", "FAILED", e)
PY
done
echo done
