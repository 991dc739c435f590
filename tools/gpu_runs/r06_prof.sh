#!/bin/bash
# round 6: the whole GPU suite, then PMC passes of msda_fwd_heads at the config-2 geometry (tools/msda_probe.py: one variant per
# process, separate --pmc passes), the bench line, and rocprofv3's kernel trace of the bench command cut per clip.  usage: r06_prof.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-v1}
O=$R/gpurun_out/r06_prof_$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
run() {  # tag, probe args
  T=$1; shift
  python tools/msda_probe.py "$@" > $O/${T}_time.json 2>/dev/null
  cat $O/${T}_time.json
  : > $O/${T}_pmc.txt
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    timeout 200 rocprofv3 --output-format csv --pmc $C -d $O/pmc_${T}_$N -o p -- python tools/msda_probe.py "$@" > $O/pmc_${T}_$N.log 2>&1
    python tools/pmc_summary.py $O/pmc_${T}_$N msda_fwd >> $O/${T}_pmc.txt 2>&1
    rm -rf $O/pmc_${T}_$N $O/pmc_${T}_$N.log
  done
  grep -v "^void\|^univs" $O/${T}_pmc.txt | tr -s ' ' | tr '\n' ';'; echo
}
run heads_default --gen 6
run heads_contig --gen 6 --cfg msda_sched=1
run heads_12x8 --gen 6 --cfg msda_strip_w=12,msda_strip_h=8
run heads_cfg5 --gen 6 --geom cfg5 --T 10
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc $?"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config5 --no-config4 --no-sliding-loop --no-frame-sharded > $O/bench_rocprof.json 2> $O/trace.err
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
CSV=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/clip_breakdown.py $CSV --skip 4 --last 8 --top 70 > $O/clip_breakdown.txt 2>&1
rm -rf $O/trace
cd $R
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("frames/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), "enqueue", round(d["host_enqueue_ms_per_step"], 2), "err", d.get("mask_logit_max_abs_err"))
print(json.dumps(d.get("roofline"))[:700])
PY
head -12 $O/kernel_stats.csv | cut -c1-150
