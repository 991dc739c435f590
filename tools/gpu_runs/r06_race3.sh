#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
echo "== two processes"; for i in 1 2; do python tools/race_probe.py --tag p$i --iters 40 2>&1 | grep -v amdgpu | cut -c1-260 & done; wait
echo "== two processes, no swin in the loop"; for i in 1 2; do python tools/race_probe.py --tag p$i --iters 40 --no-swin 2>&1 | grep -v amdgpu | cut -c1-260 & done; wait
