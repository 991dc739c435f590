#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_x
mkdir -p $O
cd $R
S=$(date +%s); timeout 115 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench wall seconds: $(( $(date +%s) - S ))" >> $O/bench.err; echo done
