#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_s
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_modules_gpu.py -q -m gpu -s -p no:cacheprovider -k "swin or config2 or b1 or config5 or config4_teacher" > $O/tests.log 2>&1
echo "pytest rc $?" >> $O/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench.json 2> $O/bench.err
echo done
