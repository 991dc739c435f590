#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_host
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python tools/prof_prompted_host.py 2>&1 | grep -v amdgpu > $O/prompted_host.txt
timeout 600 python tools/prof_prompted_host.py --first 2>&1 | grep -v amdgpu > $O/first_host.txt
head -60 $O/prompted_host.txt | cut -c1-200
