#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_e
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_graphs_gpu.py -q -m gpu -s -p no:cacheprovider -x > $O/graphs.log 2>&1
echo "pytest rc $?" >> $O/graphs.log
UNIVS_GRAPHS=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_graphs.json 2> $O/bench_graphs.err
UNIVS_GRAPHS=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_eager.json 2> $O/bench_eager.err
echo done
