#!/bin/bash
# round 5: L2 hit / miss, L1 -> L2 read requests and fabric read requests per wide-K GEMM launch: the two-dimensional tiling (linear_ablate 0) against the pass kernel (6)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_j
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
cd /tmp; export TMPDIR=/tmp
for AB in 0 6; do
  i=0
  for C in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --output-format csv --pmc $C -d $O/a${AB}p$i -o p -- python $R/tools/gemm_traffic.py --linear-ablate $AB > $O/a${AB}p$i.log 2>&1
    echo "== linear_ablate $AB pass $i" >> $O/summary.txt
    python $R/tools/gemm_traffic.py --summarise $O/a${AB}p$i >> $O/summary.txt 2>&1
    rm -rf $O/a${AB}p$i
  done
done
cat $O/a0p1.log | tail -8
cat $O/summary.txt
