#!/bin/bash
# round 4, GPU call K: fused MLP at C = 384 (one weight image, two barriers per chunk, 4-wave workgroups), src_flatten / lvl_pos
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_k
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -s -p no:cacheprovider -k "mlp_fused" > $O/ops.log 2>&1
echo "pytest rc $?" >> $O/ops.log
timeout 300 python tools/kbench.py --only mlp > $O/kbench_mlp.txt 2>&1
timeout 900 python -m pytest tests/test_modules_gpu.py -q -m gpu -p no:cacheprovider -k "swin or config2 or pixel_decoder or g2_" > $O/parity.log 2>&1
echo "pytest rc $?" >> $O/parity.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench.json 2> $O/bench.err
echo done
