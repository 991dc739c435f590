#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
echo "== alone"; python tools/race_probe.py --tag alone --iters 20 2>&1 | grep -v amdgpu
echo "== two processes"; for i in 1 2; do python tools/race_probe.py --tag p$i --iters 30 2>&1 | grep -v amdgpu & done; wait
for sw in "UNIVS_MSDA_HEADS=0" "UNIVS_FUSED_MLP=0" "UNIVS_SWIN_FUSED_LINEAR=0"; do
  echo "== two processes, $sw"; for i in 1 2; do env $sw python tools/race_probe.py --tag p$i --iters 30 2>&1 | grep -v amdgpu & done; wait
done
