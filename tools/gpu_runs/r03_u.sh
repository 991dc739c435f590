#!/bin/bash
# last profiles of round 3 (tree with the three-product window attention as the Swin default): kernel trace of the bench
# command cut per clip, the prompted steady-state clip cut per clip, then the default bench command untouched by a profiler
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_u
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench_rocprof.json 2> $O/trace.err
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
CSV=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/clip_breakdown.py $CSV --skip 4 --last 8 --top 70 > $O/clip_breakdown.txt 2>&1
rm -rf $O/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prompted -- python $R/tools/prompted_clip.py --clips 10 > $O/prompted_run.log 2>&1
CSV=$(ls $O/prompted/*/*_kernel_trace.csv | head -1)
python $R/tools/clip_breakdown.py $CSV --last 8 --top 70 > $O/prompted_clip_breakdown.txt 2>&1
rm -rf $O/prompted
cd $R
S=$(date +%s); timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench wall seconds: $(( $(date +%s) - S ))" >> $O/bench.err; echo done
