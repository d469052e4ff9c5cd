#!/bin/bash
# r07c: per-kernel durations of the small workloads, this tree against round 4's tree (same box); row-scan probe with the grouping fixed
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_rst.so timeout -k 5 300 python tools/rowscan_probe.py --workload 4k1 --mode load --steps 200 2>&1 | tail -1
OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_rst.so timeout -k 5 300 python tools/rowscan_probe.py --workload 4k1 --mode load --steps 200 2>&1 | tail -1
} < /dev/null > $O/r07c_rowscan_probe.txt 2>&1
cat $O/r07c_rowscan_probe.txt
{
for w in vga1 1080p1; do
  echo "== this tree, $w"
  bash tools/ktrace.sh gpurun_out/r07c_kt_new_$w.md --workload $w --steps 600 --warmup 100 2>&1 | grep -E "kernel \||k_blob_lds|k_mog_fused|k_rowscan|k_publish"
  echo "== round-4 tree, $w"
  ( cd build/r04_tree && bash tools/ktrace.sh gpurun_out/kt_$w.md --workload $w --steps 600 --warmup 100 2>&1 | grep -E "k_blob_lds|k_mog_fused|k_rowscan|k_publish" )
done
} < /dev/null > $O/r07c_small_workloads_ktrace.txt 2>&1
cat $O/r07c_small_workloads_ktrace.txt
