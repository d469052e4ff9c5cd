#!/bin/bash
# Back half A/B on one box: tools/backhalf_ab.sh OUT "variants"   (default = the product library; liboatgpu_<v>.so otherwise)
# 1. the blob / contour / hot-path GPU tests on the product library, 2. pipelined bench lines, 3. kernel traces of the
# pipelined 4K run (back-half kernels beside K1) and of synchronous single-frame steps (the kernels alone).
out=${1:-gpurun_out/bh}; variants=${2:-"default base"}
R=$PWD; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "contour or blob or morph or erode or hot_path or speculation or all_pass or detect or wide_frame or declined" > $out/tests.txt 2>&1; tail -3 $out/tests.txt
timeout 600 python tools/fuzz.py --configs 300 > $out/fuzz.txt 2>&1; tail -2 $out/fuzz.txt
bash tools/ab.sh $out "$variants" "--workload 4k1 --steps 1000;--workload 1080p16 --steps 200 --warmup 40;--workload 1080p1 --steps 1500" | tee $out/ab.txt
for v in $variants; do
  lib=$R/build/variants/liboatgpu_$v.so; [ "$v" = default ] && lib=$R/oat_amd/lib/liboatgpu.so
  echo "== ktrace 4k1 pipelined, variant '$v'" | tee -a $out/ab.txt
  OATGPU_MEASURE_PY=1 OATGPU_LIB=$lib bash tools/ktrace.sh $out/kt_$v.md --workload 4k1 --steps 1000 --warmup 40 | grep -E "kernel \||k_blob_lds|k_mog_fused|k_rowscan" | tee -a $out/ab.txt
  echo "== ktrace 1080p1 pipelined, variant '$v'" | tee -a $out/ab.txt
  OATGPU_MEASURE_PY=1 OATGPU_LIB=$lib bash tools/ktrace.sh $out/kt1080_$v.md --workload 1080p1 --steps 1000 --warmup 40 | grep -E "k_blob_lds|k_mog_fused|k_rowscan" | tee -a $out/ab.txt
  echo "== synchronous single-frame steps, variant '$v'" | tee -a $out/ab.txt
  OATGPU_MEASURE_PY=1 OATGPU_LIB=$lib bash tools/ktrace_latency.sh $out/lat_$v.md | grep -E "per synchronous|k_blob_lds|k_mog_fused|k_rowscan" | tee -a $out/ab.txt
done
