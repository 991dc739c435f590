#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_p
mkdir -p $O
cd $R
S=$(date +%s); timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench wall seconds: $(( $(date +%s) - S ))" >> $O/bench.err; echo done
