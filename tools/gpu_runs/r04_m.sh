#!/bin/bash
# round 4, GPU call M: decoder memory (transpose + level embedding + position embedding) in one kernel; full GPU suite, bench, timeline
# level in one Linear each (strided K / V in the kernel); full GPU suite, bench, breakdown + timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_m
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/gputests.log 2>&1
echo "pytest rc $?" >> $O/gputests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench_rocprof.json 2> $O/trace.err
CSV=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/clip_breakdown.py $CSV --skip 4 --last 8 --top 80 > $O/clip_breakdown.txt 2>&1
python $R/tools/clip_breakdown.py $CSV --skip 6 --last 1 --timeline > $O/clip_timeline.txt 2>&1
rm -rf $O/trace
echo done
