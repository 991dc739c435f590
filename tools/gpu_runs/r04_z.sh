#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_z
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 600 python tools/gpu_runs/r04_z.py > $O/z.json 2> $O/z.err
cat $O/z.json | head -40; tail -3 $O/z.err
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "linear" > $O/ops.log 2>&1; tail -2 $O/ops.log
