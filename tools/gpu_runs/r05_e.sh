#!/bin/bash
# round 5: the exact read traffic of msda_fwd_strips by REQUEST SIZE (TCC_EA0_RDREQ total / 32 B / 64 B; the rest are 128-B requests), the
# same counters on the calibration probe (known bytes), FETCH_SIZE / WRITE_SIZE of the kernel again.  Separate --pmc passes.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_e
mkdir -p $O
cd $R
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/probes/fetch_calib.hip -o /tmp/fetch_calib > $O/calib_build.log 2>&1
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum" FETCH_SIZE WRITE_SIZE; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --output-format csv --pmc $C -d $O/pmc_$N -o p -- python tools/kbench.py --only strips > $O/pmc_$N.log 2>&1
  python tools/pmc_summary.py $O/pmc_$N strips > $O/strips_$N.txt 2>&1
  rm -rf $O/pmc_$N
  cat $O/strips_$N.txt
done
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --output-format csv --pmc $C -d $O/pmcc_$N -o p -- /tmp/fetch_calib > $O/pmcc_$N.log 2>&1
  python tools/pmc_summary.py $O/pmcc_$N calib > $O/calib_$N.txt 2>&1
  rm -rf $O/pmcc_$N
  cat $O/calib_$N.txt
done
echo done
