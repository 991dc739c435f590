#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_race
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python tools/race_probe7.py --iters 30 2>&1 | grep -v amdgpu | cut -c1-700 > $O/log7_swin.txt
timeout 600 python tools/race_probe7.py --iters 30 --what head 2>&1 | grep -v amdgpu | cut -c1-700 > $O/log7_head.txt
cat $O/log7_swin.txt $O/log7_head.txt
