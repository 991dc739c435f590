#!/bin/bash
# round-3 closing run: GPU suite, smoke, the default bench command, then the rocprofv3 kernel trace of the bench command
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_w
mkdir -p $O
cd $R
timeout 280 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc $?" >> $O/smoke.log
S=$(date +%s); timeout 160 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench wall seconds: $(( $(date +%s) - S ))" >> $O/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 110 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench_rocprof.json 2> $O/trace.err
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
CSV=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/clip_breakdown.py $CSV --skip 4 --last 8 --top 70 > $O/clip_breakdown.txt 2>&1
rm -rf $O/trace
echo done
