#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_r
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "pyramid or resample" > $O/tests.log 2>&1
echo "pytest rc $?" >> $O/tests.log
timeout 900 python -m pytest tests/test_modules_gpu.py -q -m gpu -p no:cacheprovider -k "head_matches or config2 or g4_g5" > $O/tests2.log 2>&1
echo "pytest rc $?" >> $O/tests2.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench.json 2> $O/bench.err
echo done
