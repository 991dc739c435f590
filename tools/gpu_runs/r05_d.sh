#!/bin/bash
# round 5, call: the device sampler on the GPU (test), the video-loop stage profile in both sampler modes, a short bench with the new legs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_d
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_modules_gpu.py tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "device_sampler or normalize_pad or clip_loop_on_device" > $O/tests.log 2>&1
echo "pytest rc $?" >> $O/tests.log; tail -4 $O/tests.log
timeout 300 python tools/prof_video_loop.py > $O/video_loop_reference_sampler.txt 2>&1
UNIVS_SAMPLER=device timeout 300 python tools/prof_video_loop.py --cprofile 8 > $O/video_loop_device_sampler.txt 2>&1
head -14 $O/video_loop_reference_sampler.txt; head -50 $O/video_loop_device_sampler.txt
