#!/bin/bash
# round 4, GPU call F: 1 x 1 convolutions + input_proj GroupNorm through the HIP operators; pixel decoder goldens; bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_f
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -s -p no:cacheprovider -k "conv1x1 or conv3x3 or presplit" > $O/ops.log 2>&1
echo "pytest rc $?" >> $O/ops.log
timeout 900 python -m pytest tests/test_modules_gpu.py -q -m gpu -p no:cacheprovider -k "pixel_decoder or g2_ or config2 or config1 or head" > $O/parity.log 2>&1
echo "pytest rc $?" >> $O/parity.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench.json 2> $O/bench.err
echo done
