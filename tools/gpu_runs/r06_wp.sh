#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
for n in 1 4 10; do timeout 300 python tools/bench_write_prompt.py --entities $n 2>&1 | grep -v amdgpu | tail -3; done
