#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_h
mkdir -p $O
cd $R
timeout 300 python tools/prompted_clip.py --clips 10 > $O/prompted_noprof.log 2>&1
timeout 300 python tools/cprof_prompts.py > $O/prompted_cprof.txt 2>&1
timeout 300 python tools/find_syncs.py > $O/prompted_syncs.txt 2>&1
echo done
