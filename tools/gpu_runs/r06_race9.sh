#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_race
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python tools/race_probe8.py --iters 20 --set forms 2>&1 | grep -v amdgpu | cut -c1-500 > $O/log10.txt
cat $O/log10.txt
