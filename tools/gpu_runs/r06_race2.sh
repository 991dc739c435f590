#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
for sw in "X=1" "UNIVS_SPLIT_LINEAR=0" "UNIVS_SPLIT_CONV=0" "UNIVS_RESIDENT_PRESPLIT=0" "UNIVS_FUSED_NORM1=0" "UNIVS_PRESPLIT_KMIN=0" "UNIVS_FUSED_MLP=0 UNIVS_SPLIT_LINEAR=0" "UNIVS_MSDA_HEADS=0 UNIVS_MSDA_STRIPS=0"; do
  echo "== two processes, $sw"; for i in 1 2; do env $sw python tools/race_probe.py --tag p$i --iters 30 2>&1 | grep -v amdgpu | cut -c1-220 & done; wait
done
