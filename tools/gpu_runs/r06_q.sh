#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
python tools/debug_proca.py 2>&1 | grep -v amdgpu | head -12
python tools/cprof_prompts.py 2>&1 | grep -v amdgpu | head -60
