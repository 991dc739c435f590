#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
for n in 10 4; do timeout 300 python tools/bench_write_prompt_pieces.py $n 2>&1 | grep -v amdgpu; done
