#!/bin/bash
# round 5: what bounds gemm_f16x3_tile -- the same launches without the split / without the matrix instructions (instrumented builds)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_m
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "tile_kernel" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for L in "" nosplit nomfma nosplit_nomfma; do
  if [ -n "$L" ]; then export UNIVS_HIP_LIB=$R/univs_amd/libunivs_hip_$L.so; fi
  timeout 300 python tools/gemm_tile_sweep.py --quick >> $O/sweep.txt 2>> $O/sweep.err
done
cat $O/sweep.txt
