#!/bin/bash
# usage: tools/prof_msda_traffic.sh <outdir>   (run on the GPU box)
# HBM-side bytes per launch of the MSDA kernels, as the microarch guide prescribes: FETCH_SIZE and WRITE_SIZE in separate
# --pmc passes (they do not fit one pass), no trace domains in the same run.  Prints the per-dispatch means per kernel.
set -u
OUT=${1:-gpurun_out/prof_msda_traffic}
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python tools/kbench.py --only msda"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/f -o p -- $CMD > $OUT/f.log 2>&1; python tools/pmc_summary.py $OUT/f msda > $OUT/fetch.txt 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/w -o p -- $CMD > $OUT/w.log 2>&1; python tools/pmc_summary.py $OUT/w msda > $OUT/write.txt 2>&1
rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/h -o p -- $CMD > $OUT/h.log 2>&1; python tools/pmc_summary.py $OUT/h msda > $OUT/l2.txt 2>&1
cat $OUT/fetch.txt $OUT/write.txt $OUT/l2.txt
rm -rf $OUT/f $OUT/w $OUT/h
