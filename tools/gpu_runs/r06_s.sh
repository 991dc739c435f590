#!/bin/bash
# round 6: the next segment's cold window warmed into L2 during the last item of a segment: parity, timing against the build without it
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_modules_gpu.py -m gpu -q -k "heads or aten_operator" 2>&1 | tail -3
for i in 1 2 3; do
python tools/msda_probe.py --gen 6 2>/dev/null
UNIVS_HIP_LIB=$R/univs_amd/libunivs_hip_heads_noprefetch.so python tools/msda_probe.py --gen 6 2>/dev/null
done
python tools/msda_probe.py --gen 6 --geom cfg5 --T 10 2>/dev/null
UNIVS_HIP_LIB=$R/univs_amd/libunivs_hip_heads_noprefetch.so python tools/msda_probe.py --gen 6 --geom cfg5 --T 10 2>/dev/null
UNIVS_HIP_LIB=$R/univs_amd/libunivs_hip_heads_trace.so python tools/msda_trace6.py 2>/dev/null | grep -E "workgroups 256|span by|items traced"
python tools/kbench.py --only strips 2>/dev/null | grep heads
