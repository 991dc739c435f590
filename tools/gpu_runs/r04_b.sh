#!/bin/bash
# round 4, GPU call B: fused MLP v2 (fragment ring of three, static weight-stream schedule, 4-wave workgroups for C = 96) with
# ablation timings, window attention range scaling, then a rocprofv3 clip breakdown of the bench.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_b
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -s -p no:cacheprovider -k "mlp_fused or window_attention or linear_f16x3_row or linear_fused" > $O/ops.log 2>&1
echo "pytest rc $?" >> $O/ops.log
timeout 300 python tools/kbench.py --only mlp > $O/kbench_mlp.txt 2>&1
timeout 300 python tools/kbench.py --only win > $O/kbench_win.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 > $O/bench.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench_rocprof.json 2> $O/trace.err
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
CSV=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/clip_breakdown.py $CSV --skip 4 --last 8 --top 80 > $O/clip_breakdown.txt 2>&1
rm -rf $O/trace
echo done
