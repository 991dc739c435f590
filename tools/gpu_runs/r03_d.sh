#!/bin/bash
# strips iteration: parity + kbench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_d
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -s -p no:cacheprovider -x -k "strips" > $O/ops.log 2>&1
echo "pytest rc $?" >> $O/ops.log
timeout 300 python tools/kbench.py --only msda 2>/dev/null | grep -E "strips|fused3_v1|tiled2" > $O/kbench_msda.txt
echo done
