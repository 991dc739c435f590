#!/bin/bash
# round 6: generation 6 of the MSDA kernel (msda_heads.hip): parity tests, then the kernel benchmark beside generation 5
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_b
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "heads or blocked" -s > $O/pytest_heads.log 2>&1
tail -25 $O/pytest_heads.log
timeout 300 python tools/kbench.py --only msda > $O/kbench_msda.json 2> $O/kbench_msda.err
cat $O/kbench_msda.json; tail -5 $O/kbench_msda.err
