#!/bin/bash
# round 5: only the rocprofv3 kernel trace of the short bench command and its per-clip breakdown (a re-run of that part of r05_prof.sh:
# one 18.6-ms outlier launch of the encoder FFN polluted the mean of r05_prof_v4's eight clips).  usage: r05_prof_trace.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-v4b}
O=$R/gpurun_out/r05_prof_$TAG
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config5 --no-config4 --no-sliding-loop --no-frame-sharded > $O/bench_rocprof.json 2> $O/trace.err
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
CSV=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/clip_breakdown.py $CSV --skip 4 --last 8 --top 70 > $O/clip_breakdown.txt 2>&1
python $R/tools/clip_breakdown.py $CSV --skip 6 --last 1 --timeline > $O/clip_timeline.txt 2>&1
rm -rf $O/trace
head -12 $O/clip_breakdown.txt | cut -c1-120
