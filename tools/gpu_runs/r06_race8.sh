#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_race
mkdir -p $O
cd $R
export TMPDIR=/tmp
: > $O/log9.txt
timeout 600 python tools/race_probe8.py --iters 30 --set micro --variants 0 2>&1 | grep -v amdgpu | cut -c1-500 >> $O/log9.txt
timeout 600 python tools/race_probe8.py --iters 30 --set gemms --variants 0 2>&1 | grep -v amdgpu | cut -c1-500 >> $O/log9.txt
for a in nosplit nomfma nosplit_nomfma; do
  UNIVS_HIP_LIB=$R/univs_amd/libunivs_hip_$a.so timeout 600 python tools/race_probe8.py --iters 30 --set tile --variants 0 2>&1 | grep -v amdgpu | cut -c1-500 >> $O/log9.txt
done
cat $O/log9.txt
