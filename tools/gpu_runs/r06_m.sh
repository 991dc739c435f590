#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
python tools/debug_loop_determinism.py fused_cross_attention small_linear small_mlp_chain fused_mlp 2>&1 | grep -v amdgpu.ids | tail -20
