#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_i
mkdir -p $O
cd $R
for v in default gs256 default gs256; do
  if [ $v = default ]; then L=""; else L="$R/univs_amd/libunivs_hip_$v.so"; fi
  UNIVS_HIP_LIB=$L timeout 600 python tools/gemmset.py --tag $v 2>/dev/null | grep -E "s3_proj|s3_fc2|s4_|merge|conv|enc_ffn|TOTAL" | awk '{print $1, $3, $4, $11, $12}' | tr -d '",' | paste -sd' ' | cut -c1-700
done
