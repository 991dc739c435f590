#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_loop
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python tools/prof_video_loop.py --cprofile 6 2>&1 | grep -v amdgpu | cut -c1-200 > $O/predictor_cprofile.txt; sed -n 1,75p $O/predictor_cprofile.txt
