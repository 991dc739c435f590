#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_r
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_modules_gpu.py -m gpu -q -s -k "proca or aten_operator" 2>&1 | grep -E "passed|failed|ATen|Error" | head
UNIVS_HIP_LIB=$R/univs_amd/libunivs_hip_heads_trace.so python tools/msda_trace6.py > $O/trace_default.txt 2>/dev/null
cat $O/trace_default.txt
