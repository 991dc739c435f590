#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_fix
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_distributed_gpu.py -q -m gpu -x --durations=5 > $O/pytest_dist_gpu.log 2>&1; echo "rc $?"; tail -25 $O/pytest_dist_gpu.log
