#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
( timeout 600 python tools/fuzz.py --configs 302 --seed 11 --only 301 2>&1 | grep -v amdgpu.ids | tail -2 ) < /dev/null > $O/r07_fuzz_after_fix.txt 2>&1
for seed in 11 12 13; do ( timeout -k 5 1500 python tools/fuzz.py --configs 1500 --seed $seed 2>&1 | grep -v amdgpu.ids | tail -3 ) < /dev/null >> $O/r07_fuzz_after_fix.txt 2>&1; done
cat $O/r07_fuzz_after_fix.txt
( timeout -k 5 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | cut -c1-300 ) < /dev/null > $O/r07_gputests_after_fix.txt 2>&1
cat $O/r07_gputests_after_fix.txt
