#!/bin/bash
# round 6, final tree: the whole GPU suite, the bench line, rocprofv3's kernel trace of the bench command cut per clip, the video loop's stages
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-v4}
O=$R/gpurun_out/r06_final_$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "gpu suite rc $?"; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc $?"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config5 --no-config4 --no-sliding-loop --no-frame-sharded > $O/bench_rocprof.json 2> $O/trace.err
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
CSV=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/clip_breakdown.py $CSV --skip 4 --last 8 --top 70 > $O/clip_breakdown.txt 2>&1
rm -rf $O/trace
cd $R
timeout 600 python tools/prof_video_loop.py 2>&1 | grep -v amdgpu | cut -c1-220 > $O/video_loop_stages.txt
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("frames/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), "enqueue", round(d["host_enqueue_ms_per_step"], 2), "err", d.get("mask_logit_max_abs_err"), "flips", d.get("mask_sign_flips"))
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "steady", d["steady_state_with_prompts"]["ms_per_clip"])
print("loop", d["sliding_clip_loop"]["one_window"]["ms_per_video"], "cfg4", d["config4_swinb_refvos"]["ms_per_clip"], "cfg5", d["config5_swinl_1080p"]["window_attention_f16x3"]["ms_per_clip"])
PY
head -6 $O/clip_breakdown.txt | cut -c1-150; head -12 $O/video_loop_stages.txt
