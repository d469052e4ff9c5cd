#!/bin/bash
# r08st: stress of the tree of round 6 (all scratch sets at create, the no-park flag): 3 x 1 500 fuzz configurations, the soaks again
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for seed in 31 32 33; do ( timeout -k 5 1500 python tools/fuzz.py --configs 1500 --seed $seed 2>&1 | grep -v amdgpu.ids | tail -2 ); done
echo "== 3 streams 320x240, 20 000 frames, position filter off (paired back half)"
timeout -k 5 600 python tools/soak.py --frames 20000 --no-kalman 2>&1 | grep -v amdgpu.ids | tail -1
echo "== position filter on"
timeout -k 5 600 python tools/soak.py --frames 20000 2>&1 | grep -v amdgpu.ids | tail -1
echo "== one 1080p stream, 6 000 frames (paired)"
timeout -k 5 900 python tools/soak.py --frames 6000 --rows 1080 --cols 1920 --streams 1 --ring 8 --no-kalman --threads 64 2>&1 | grep -v amdgpu.ids | tail -1
echo "== two 1080p streams, 4 000 frames (early order)"
timeout -k 5 900 python tools/soak.py --frames 4000 --rows 1080 --cols 1920 --streams 2 --ring 8 --no-kalman --threads 64 2>&1 | grep -v amdgpu.ids | tail -1
echo "== one 4K stream, 1 500 frames (early order)"
timeout -k 5 900 python tools/soak.py --frames 1500 --rows 2160 --cols 3840 --streams 1 --ring 8 --no-kalman --threads 64 2>&1 | grep -v amdgpu.ids | tail -1
} < /dev/null > $O/r08st_fuzz_and_soak.txt 2>&1
cat $O/r08st_fuzz_and_soak.txt
