#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_h
mkdir -p $O
cd $R
timeout 300 python tools/prompted_clip.py --clips 10 > $O/prompted_noprof.log 2>&1
timeout 900 python -m pytest tests/test_modules_gpu.py -q -m gpu -s -p no:cacheprovider -x -k "head_matches or prefetch or config2 or g4_g5" > $O/tests.log 2>&1
echo "pytest rc $?" >> $O/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench.json 2> $O/bench.err
echo done
