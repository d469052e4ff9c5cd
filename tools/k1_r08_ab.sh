#!/bin/bash
# round 6, VERDICT r05 next-2 (i) / (iii): packed fp32 and register initialisation A/B + the dynamic instruction profile
out=${1:-gpurun_out/r08k}; mkdir -p $out
for r in 1 2 3; do
  bash tools/ab.sh $out/r$r "meas pk undef pkundef" "--workload 4k1 --steps 1000;--workload 1080p16 --steps 200 --warmup 40"
done 2>&1 | tee $out/ab.txt
bash tools/cut_profile.sh $out/cut > /dev/null 2>&1
cat $out/cut/cut.md
