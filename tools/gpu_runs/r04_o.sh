#!/bin/bash
# round 4, GPU call O: wide feature passes (192 / 256 per pass) of the streamed three-product kernel: convolutions, Swin stage 3 / 4,
# encoder / decoder projections
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_o
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 900 python tools/kbench.py --only wide > $O/kbench_wide.txt 2> $O/kbench_wide.err
echo "kbench rc $?"
tail -c 3000 $O/kbench_wide.txt
