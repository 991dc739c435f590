#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_o
mkdir -p $O
cd $R
export TMPDIR=/tmp
python tools/launch_sources.py > $O/launch_sources_prompted.txt 2> $O/err.txt
head -80 $O/launch_sources_prompted.txt | cut -c1-230; tail -3 $O/err.txt
