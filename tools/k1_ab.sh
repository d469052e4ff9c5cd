#!/bin/bash
# K1 A/B on one box: tools/k1_ab.sh OUT "variants" [reps]   -- the product library against liboatgpu_<v>.so, interleaved
out=${1:-gpurun_out/k1ab}; variants=${2:-"default base"}; reps=${3:-2}
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_zz_long_run_gpu.py -x -q -m gpu -k "two_frames or full_size or determinism or long_run or hot_path or single_launch or fusion or mog2_mask" > $out/tests.txt 2>&1; tail -3 $out/tests.txt
for r in $(seq $reps); do
  bash tools/ab.sh $out/r$r "$variants" "--workload 4k1 --steps 1000;--workload 1080p16 --steps 200 --warmup 40;--workload 1080p1 --steps 1500"
done | tee $out/ab.txt
