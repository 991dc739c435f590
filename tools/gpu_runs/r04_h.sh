#!/bin/bash
# round 4, GPU call H: small-M Linear sweep (Swin stage 3 / 4 shapes) over output features per pass and workgroups along the rows
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_h
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python tools/kbench.py --only smallm > $O/kbench_smallm.txt 2>&1
echo done
