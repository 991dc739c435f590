#!/bin/bash
# the sharded sliding loop at N = 2 on one GPU with every row of the clip's mask logits replicated (the form before ClipMaskRows): byte counters only
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_n2
mkdir -p $O
cd $R
export TMPDIR=/tmp UNIVS_BENCH_ONE_GPU_DEBUG=1 UNIVS_REPLICATE_CLIP_MASKS=1
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 2 --warmup 1 --no-frame-sharded > $O/bench_n2_repl.json 2> $O/bench_n2_repl.err
python - <<PY
import json
r = json.loads(open("gpurun_out/r06_n2/bench_n2_repl.json").read().strip().splitlines()[-1])
print(json.dumps({k: v for k, v in r["sliding_clip_loop"]["frame_sharded"].items() if "bytes" in k}))
PY
