#!/bin/bash
# round 5, call 1: (1) the cross-attention fix through its op tests; (2) the GEMM launches of a clip under the
# default library and the three instrumented builds (what the operand split / the MFMAs cost in place); (3) FETCH_SIZE / WRITE_SIZE
# calibrated on known byte counts (tools/probes/fetch_calib.hip), separate --pmc passes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_a
mkdir -p $O
cd $R
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "cross_attention or deferred" -s > $O/xattn.log 2>&1
echo "pytest rc $?" >> $O/xattn.log
grep -E "cross attention|passed|failed|rc" $O/xattn.log | tail -30
for v in default nosplit nomfma nosplit_nomfma; do
  if [ $v = default ]; then L=""; else L="$R/univs_amd/libunivs_hip_$v.so"; fi
  UNIVS_HIP_LIB=$L timeout 600 python tools/gemmset.py --tag $v > $O/gemmset_$v.txt 2> $O/gemmset_$v.err
  tail -1 $O/gemmset_$v.txt
done
hipcc --offload-arch=gfx950 -O3 tools/probes/fetch_calib.hip -o /tmp/fetch_calib > $O/calib_build.log 2>&1
timeout 300 /tmp/fetch_calib > $O/calib_times.txt 2>&1; cat $O/calib_times.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --output-format csv --pmc $c -d $O/pmc_$n -o p -- /tmp/fetch_calib > $O/pmc_$n.log 2>&1
  python tools/pmc_summary.py $O/pmc_$n calib > $O/calib_$n.txt 2>&1
  rm -rf $O/pmc_$n
  cat $O/calib_$n.txt
done
echo done
