#!/bin/bash
# round 5: tile-shape / load-depth sweep of gemm_f16x3_tile on the clip's wide-K Linears
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_l
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "tile_kernel" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python tools/gemm_tile_sweep.py > $O/sweep.txt 2> $O/sweep.err; cat $O/sweep.txt; tail -3 $O/sweep.err
