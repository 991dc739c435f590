#!/bin/bash
# round 3, GPU call B: generation-5 MSDA (strips, head-major operands): operator parity, kbench, module parity at full size
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_b
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -s -p no:cacheprovider -x -k "strips or linear_blocked" > $O/ops.log 2>&1
echo "pytest rc $?" >> $O/ops.log
timeout 300 python tools/kbench.py --only msda > $O/kbench_msda.txt 2>&1
timeout 600 python -m pytest tests/test_modules_gpu.py -q -m gpu -s -p no:cacheprovider -k "config2 or pixel_decoder or g2_" > $O/modules.log 2>&1
echo "pytest rc $?" >> $O/modules.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
echo done
