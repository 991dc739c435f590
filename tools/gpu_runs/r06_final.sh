#!/bin/bash
# round 6, final tree: the stand-alone reproduction of hazard 23, the bench line, rocprofv3's kernel trace of the bench command cut per clip.
# usage: r06_final.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-v3}
O=$R/gpurun_out/r06_final_$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/cohab_repro.hip -o /tmp/cohab_repro 2> $O/repro_build.err && timeout 300 /tmp/cohab_repro > $O/cohab_repro.txt 2>&1
echo "repro rc $?"; cat $O/cohab_repro.txt
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
print("frames/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), "enqueue", round(d["host_enqueue_ms_per_step"], 2), "err", d.get("mask_logit_max_abs_err"), "flips", d.get("mask_sign_flips"))
print(json.dumps(d.get("roofline"))[:400])
print("steady", json.dumps(d.get("steady_state_with_prompts"))[:300])
PY
head -8 $O/clip_breakdown.txt | cut -c1-150
