#!/bin/bash
# round 5: the whole GPU suite (what the driver runs at round end), smoke(), then the bench line.  usage: r05_full.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-a}
shift
O=$R/gpurun_out/r05_full_$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/gpu_tests.log 2>&1
echo "pytest rc $?" >> $O/gpu_tests.log
tail -15 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py "$@" > $O/bench.json 2> $O/bench.err
echo "bench rc $?"
python - <<PY
import json
try:
    d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    print("frames/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), "enqueue", round(d["host_enqueue_ms_per_step"], 2), "err", d.get("mask_logit_max_abs_err"))
    for k in ("roofline", "steady_state_with_prompts", "config4_swinb_refvos", "sliding_clip_loop", "frame_sharded_n1"):
        print(k, json.dumps(d.get(k))[:700])
except Exception as e:
    print("bench parse failed", e)
PY
tail -5 $O/bench.err
