#!/bin/bash
# r07p: soaks and fuzz of the shipped tree (paired back half, four-set early order), then tests/test_gpu_multirank.py 20 times without a retry
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
echo "== paired back half, small frames, position filter off (3 streams 320x240, 20 000 frames)"
timeout -k 5 600 python tools/soak.py --frames 20000 --no-kalman 2>&1 | grep -v amdgpu.ids | tail -2
echo "== position filter on (full launch sequence: plain order)"
timeout -k 5 600 python tools/soak.py --frames 20000 2>&1 | grep -v amdgpu.ids | tail -2
echo "== one 1080p stream, paired back half (6 000 frames)"
timeout -k 5 900 python tools/soak.py --frames 6000 --rows 1080 --cols 1920 --streams 1 --ring 8 --no-kalman --threads 32 2>&1 | grep -v amdgpu.ids | tail -2
echo "== two 1080p streams: early order, four scratch sets (4 000 frames)"
timeout -k 5 900 python tools/soak.py --frames 4000 --rows 1080 --cols 1920 --streams 2 --ring 8 --no-kalman --threads 32 2>&1 | grep -v amdgpu.ids | tail -2
echo "== three 1080p streams: early order (3 000 frames)"
timeout -k 5 900 python tools/soak.py --frames 3000 --rows 1080 --cols 1920 --streams 3 --ring 8 --no-kalman --threads 32 2>&1 | grep -v amdgpu.ids | tail -2
echo "== one 4K stream: early order (1 500 frames)"
timeout -k 5 900 python tools/soak.py --frames 1500 --rows 2160 --cols 3840 --streams 1 --ring 8 --no-kalman --threads 32 2>&1 | grep -v amdgpu.ids | tail -2
} < /dev/null > $O/r07p_soak.txt 2>&1
cat $O/r07p_soak.txt
( timeout -k 5 900 python tools/fuzz.py --configs 400 --seed 7 2>&1 | grep -v amdgpu.ids | tail -3 ) < /dev/null > $O/r07p_fuzz.txt 2>&1
cat $O/r07p_fuzz.txt
{
echo "# tests/test_gpu_multirank.py, 20 consecutive runs, no second attempt anywhere (VERDICT r04 item 7); a failing launch leaves gpurun_out/multirank_failure_*.txt"
for i in $(seq 1 20); do
  t0=$(date +%s)
  r=$(timeout -k 5 1200 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | tail -1)
  echo "run $i: $r  ($(( $(date +%s) - t0 )) s wall)"
done
} < /dev/null > $O/r07p_multirank_20x.txt 2>&1
cat $O/r07p_multirank_20x.txt
ls $O | grep multirank_failure
