#!/bin/bash
# round 4, GPU call A: the fused MLP kernel (csrc/mlp_f16x3.hip) -- operator parity, kernel timings against the two fused Linears,
# the model goldens that run through it (g2 / g3 / g9 / cfg 2), a bench line, and the baseline of the same tree with the fusion off.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_a
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -s -p no:cacheprovider -k "mlp_fused" > $O/ops.log 2>&1
echo "pytest rc $?" >> $O/ops.log
timeout 300 python tools/kbench.py --only mlp > $O/kbench_mlp.txt 2>&1
timeout 900 python -m pytest tests/test_modules_gpu.py -q -m gpu -p no:cacheprovider -k "g2_ or pixel_decoder or swin or config2" > $O/parity.log 2>&1
echo "pytest rc $?" >> $O/parity.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 > $O/bench_fused.json 2> $O/bench_fused.err
UNIVS_FUSED_MLP=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 > $O/bench_unfused.json 2> $O/bench_unfused.err
echo done
