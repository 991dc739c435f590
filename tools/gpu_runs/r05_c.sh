#!/bin/bash
# round 5, call 3: the W-resident Linear on the split image: bit-identity tests, the Linear tests, gemmset with the switch off / on,
# phase trace of the new staging
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_c
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "linear" > $O/linear_tests.log 2>&1
echo "pytest rc $?" >> $O/linear_tests.log
tail -5 $O/linear_tests.log
for v in 0 1; do
  UNIVS_RESIDENT_PRESPLIT=$v timeout 600 python tools/gemmset.py --tag presplit$v > $O/gemmset_presplit$v.txt 2> $O/gemmset_presplit$v.err
  tail -1 $O/gemmset_presplit$v.txt
done
UNIVS_HIP_LIB=$R/univs_amd/libunivs_hip_trace.so timeout 300 python tools/gemm_trace.py > $O/gemm_trace.txt 2> $O/gemm_trace.err
grep -A3 "s3_qkv\|enc_value" $O/gemm_trace.txt | cut -c1-200
echo done
