#!/bin/bash
# round 5, call 2: in-kernel phase timelines of the GEMM kernels (instrumented builds), counters available for read-request sizes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_b
mkdir -p $O
cd $R
export TMPDIR=/tmp
for v in trace trace_nosplit_nomfma; do
  UNIVS_HIP_LIB=$R/univs_amd/libunivs_hip_$v.so timeout 300 python tools/gemm_trace.py > $O/gemm_trace_$v.txt 2> $O/gemm_trace_$v.err
  tail -3 $O/gemm_trace_$v.err
done
rocprofv3 -L > $O/counters.txt 2>&1
grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_[A-Z_]*128B[A-Z_]*\|TCC_BUBBLE[A-Z_]*" $O/counters.txt | sort -u | head -60
echo done
