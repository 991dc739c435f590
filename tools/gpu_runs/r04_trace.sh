#!/bin/bash
# round 4: steady-state clip breakdown (rocprofv3 kernel trace of the bench command) of the current tree; $1 = tag, rest = env assignments
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-t}
shift
O=$R/gpurun_out/r04_trace_$TAG
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
cd /tmp; export TMPDIR=/tmp
env "$@" timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench_rocprof.json 2> $O/trace.err
CSV=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/clip_breakdown.py $CSV --skip 4 --last 8 --top 70 > $O/clip_breakdown.txt 2>&1
python $R/tools/clip_breakdown.py $CSV --skip 6 --last 1 --timeline > $O/clip_timeline.txt 2>&1
rm -rf $O/trace
head -45 $O/clip_breakdown.txt | cut -c1-150
