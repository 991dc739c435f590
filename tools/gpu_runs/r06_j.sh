#!/bin/bash
# round 6: the GPU suite again (stores of repeated query slots masked), then the PMC passes of generation 6 as the model runs it
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_j
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -8 $O/pytest_gpu.log
