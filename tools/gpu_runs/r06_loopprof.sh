#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_loop
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python tools/prof_video_loop.py --cprofile-post detect_newly_entities_per_clip_instance 2>&1 | grep -v amdgpu | cut -c1-220 > $O/detect.txt
timeout 600 python tools/prof_video_loop.py --cprofile-post write_prompt_predictions_into_annotations_per_clip 2>&1 | grep -v amdgpu | cut -c1-220 > $O/write_prompt.txt
head -70 $O/detect.txt
