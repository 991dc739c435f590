#!/bin/bash
# round 5: GEMM kernels after the vector-memory wait fixes (branch-free W stream with a full group of lead, batched residual loads): parity subset + timings
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-k}
O=$R/gpurun_out/r05_$TAG
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "linear or mlp or conv or gemm or stream or ffn or presplit" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 600 python tools/gemmset.py --tag $TAG > $O/gemmset.json 2> $O/gemmset.err
tail -40 $O/gemmset.json | cut -c1-200
