#!/bin/bash
# round 5: the bench line (the driver's command), rocprofv3 kernel trace of the bench command cut per clip, the same for the prompted
# steady-state clip and for the config-4 clip.  usage: r05_prof.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-v1}
O=$R/gpurun_out/r05_prof_$TAG
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc $?"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config5 --no-config4 --no-sliding-loop --no-frame-sharded > $O/bench_rocprof.json 2> $O/trace.err
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
CSV=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/clip_breakdown.py $CSV --skip 4 --last 8 --top 70 > $O/clip_breakdown.txt 2>&1
python $R/tools/clip_breakdown.py $CSV --skip 6 --last 1 --timeline > $O/clip_timeline.txt 2>&1
rm -rf $O/trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tracep -o t -- python $R/tools/prompted_clip.py --clips 8 > $O/prompted.log 2> $O/tracep.err
CSV=$(find $O/tracep -name "*kernel_trace.csv" | head -1)
python $R/tools/clip_breakdown.py $CSV --last 6 --top 70 > $O/prompted_clip_breakdown.txt 2>&1
python $R/tools/clip_breakdown.py $CSV --last 1 --timeline > $O/prompted_clip_timeline.txt 2>&1
rm -rf $O/tracep
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace4 -o t -- python $R/tools/cfg4_clip.py --clips 6 > $O/cfg4.log 2> $O/trace4.err
CSV=$(find $O/trace4 -name "*kernel_trace.csv" | head -1)
python $R/tools/clip_breakdown.py $CSV --skip 3 --last 4 --top 70 > $O/cfg4_clip_breakdown.txt 2>&1
rm -rf $O/trace4
cd $R
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("frames/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), "enqueue", round(d["host_enqueue_ms_per_step"], 2), "err", d.get("mask_logit_max_abs_err"))
for k in ("steady_state_with_prompts", "config4_swinb_refvos", "sliding_clip_loop", "cpu_baseline"):
    print(k, json.dumps(d.get(k))[:900])
PY
head -30 $O/clip_breakdown.txt | cut -c1-140; cat $O/prompted.log $O/cfg4.log | grep -v amdgpu; head -8 $O/prompted_clip_breakdown.txt | cut -c1-140; head -8 $O/cfg4_clip_breakdown.txt | cut -c1-140
