#!/bin/bash
# round 4, GPU call P: the Linear in front of the fused MLP in the same kernel (encoder tail, Swin block tail): op tests, module
# goldens, bench with and without; A-fragment batches of the streamed kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_p
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -s -p no:cacheprovider -k "proj_mlp or mlp_fused" > $O/ops.log 2>&1
echo "pytest rc $?" >> $O/ops.log
grep -E "proj_mlp_fused|passed|failed|rc " $O/ops.log | tail -14
timeout 900 python -m pytest tests/test_modules_gpu.py -q -m gpu -p no:cacheprovider -k "swin or pixel_decoder or g2 or config2 or autocast or config4" > $O/parity.log 2>&1
echo "pytest rc $?" >> $O/parity.log
tail -4 $O/parity.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench.json 2> $O/bench.err
UNIVS_FUSED_PROJ_MLP=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench_off.json 2> $O/bench_off.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench2.json 2> $O/bench2.err
for f in bench bench_off bench2; do python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1]); print("$f", round(d["value"],1), round(d["ms_per_step"],3), d["host_enqueue_ms_per_step"], d.get("mask_logit_max_abs_err"))
except Exception as e: print("$f", "FAILED", e)
PY
done
timeout 600 python tools/kbench.py --only batches > $O/kbench_batches.txt 2> $O/kbench_batches.err
grep batches_ $O/kbench_batches.txt | cut -c1-220
echo done
