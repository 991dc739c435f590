#!/bin/bash
# kernel trace of the bench command, cut per clip
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_l
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench_rocprof.json 2> $O/trace.err
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
CSV=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/clip_breakdown.py $CSV --skip 4 --last 8 --top 70 > $O/clip_breakdown.txt 2>&1
rm -rf $O/trace
echo done
