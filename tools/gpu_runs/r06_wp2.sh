#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_loop
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mask_stats_gpu.py -q -m gpu -x > $O/pytest_maskstats.log 2>&1; echo "mask_stats rc $?"; tail -4 $O/pytest_maskstats.log
timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_vos_gpu.py -q -m gpu -x -k "loop or long_video or config3 or vos" > $O/pytest_loop.log 2>&1; echo "loop tests rc $?"; tail -3 $O/pytest_loop.log
for n in 1 4 10; do timeout 300 python tools/bench_write_prompt.py --entities $n 2>&1 | grep -v amdgpu | tail -1; done
timeout 600 python tools/prof_video_loop.py 2>&1 | grep -v amdgpu | cut -c1-200 | head -11
