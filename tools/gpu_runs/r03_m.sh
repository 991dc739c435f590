#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_m
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -s -p no:cacheprovider -k "window" > $O/tests.log 2>&1
echo "pytest rc $?" >> $O/tests.log
echo done
