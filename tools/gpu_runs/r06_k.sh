#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_k
mkdir -p $O
cd $R
export TMPDIR=/tmp
python tools/msda_determinism.py 2>&1 | grep -v amdgpu.ids | tee $O/determinism.txt
UNIVS_MSDA_HEADS=0 timeout 600 python -m pytest tests/test_modules_gpu.py -m gpu -q -k "device_sampler" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_modules_gpu.py -m gpu -q -k "device_sampler" 2>&1 | tail -3
