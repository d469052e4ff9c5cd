#!/bin/bash
# r07s: 20 000-frame soaks of the 2- and 3-stream defaults (early order, four scratch sets, one wave a K1 workgroup) and of one 1080p stream (paired back half)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
echo "== two 1080p streams: early order (20 000 frames)"
timeout -k 5 1500 python tools/soak.py --frames 20000 --rows 1080 --cols 1920 --streams 2 --ring 8 --no-kalman --threads 64 2>&1 | grep -v amdgpu.ids | tail -2
echo "== three 1080p streams: early order (20 000 frames)"
timeout -k 5 1800 python tools/soak.py --frames 20000 --rows 1080 --cols 1920 --streams 3 --ring 8 --no-kalman --threads 64 2>&1 | grep -v amdgpu.ids | tail -2
echo "== one 1080p stream: paired back half (20 000 frames)"
timeout -k 5 1200 python tools/soak.py --frames 20000 --rows 1080 --cols 1920 --streams 1 --ring 8 --no-kalman --threads 64 2>&1 | grep -v amdgpu.ids | tail -2
} < /dev/null > $O/r07s_soak_20000.txt 2>&1
cat $O/r07s_soak_20000.txt
