#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
python tools/kbench.py --only mlp 2>/dev/null | grep -E "encoder_ffn"
