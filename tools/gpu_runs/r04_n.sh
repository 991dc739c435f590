#!/bin/bash
# round 4, GPU call N: PatchEmbed kernel; Swin goldens, cfg 2, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_n
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -s -p no:cacheprovider -k "patch_embed or decoder_memory" > $O/ops.log 2>&1
echo "pytest rc $?" >> $O/ops.log
timeout 900 python -m pytest tests/test_modules_gpu.py -q -m gpu -p no:cacheprovider -k "swin or config2 or config5 or autocast" > $O/parity.log 2>&1
echo "pytest rc $?" >> $O/parity.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench2.json 2> $O/bench2.err
echo done
