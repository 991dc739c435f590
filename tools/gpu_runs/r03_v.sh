#!/bin/bash
# mask-family experiment: one-shot exact-f32 kernel for the coarse attention masks, workgroup count of the split kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_v
mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "mask_decode or presplit or configure" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
timeout 150 python tools/kbench.py --only mask > $O/kbench_mask.txt 2>&1
echo done
