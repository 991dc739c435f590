#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_j
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -s -p no:cacheprovider -k "linear or conv3x3 or presplit" > $O/tests.log 2>&1
echo "pytest rc $?" >> $O/tests.log

echo done
