#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_n
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "linear" > $O/tests.log 2>&1
echo "pytest rc $?" >> $O/tests.log
timeout 900 python tools/kbench.py --only swinlin > $O/kbench.json 2> $O/kbench.err
echo done
