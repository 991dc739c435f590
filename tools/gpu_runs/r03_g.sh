#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_g
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -s -p no:cacheprovider -k "window" > $O/win.log 2>&1
echo "pytest rc $?" >> $O/win.log
timeout 900 python -m pytest tests/test_modules_gpu.py -q -m gpu -s -p no:cacheprovider -k "config5_fp16 or config5_swinl" > $O/cfg5.log 2>&1
echo "pytest rc $?" >> $O/cfg5.log
timeout 600 python tools/kbench.py --only win12 > $O/kbench_win12.json 2> $O/kbench_win12.err
echo done
