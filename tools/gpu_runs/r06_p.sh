#!/bin/bash
# round 6: ProCA without building `memory` (csrc/proca_attn.hip) + cached frequency vectors: parity, launch sources, prompted clip time
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_p
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_modules_gpu.py -m gpu -q -k "proca or head_matches or clip_loop_on_device or config3_long_video_on_device" > $O/pytest.log 2>&1
tail -4 $O/pytest.log
python tools/launch_sources.py 2>/dev/null | head -30
python tools/prompted_clip.py --clips 8 2>/dev/null | tail -5
UNIVS_FUSED_PROCA=0 python tools/prompted_clip.py --clips 8 2>/dev/null | tail -2
