#!/bin/bash
# round 4, GPU call G: cross-attention with prefetch + Q in LDS, decoder launch cuts (MLP ReLU in the GEMM epilogue, q/k in one
# projection, heads on contiguous tokens), PatchMerging reduction on the three-product Linear; tests, bench, breakdown, timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_g
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -s -p no:cacheprovider -k "cross_attention" > $O/ops.log 2>&1
echo "pytest rc $?" >> $O/ops.log
timeout 1200 python -m pytest tests/test_modules_gpu.py -q -m gpu -p no:cacheprovider > $O/parity.log 2>&1
echo "pytest rc $?" >> $O/parity.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench_rocprof.json 2> $O/trace.err
CSV=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/clip_breakdown.py $CSV --skip 4 --last 8 --top 80 > $O/clip_breakdown.txt 2>&1
python $R/tools/clip_breakdown.py $CSV --skip 6 --last 1 --timeline > $O/clip_timeline.txt 2>&1
rm -rf $O/trace
echo done
