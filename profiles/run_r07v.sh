#!/bin/bash
# r07v: the library checks that its four streams run side by side and replaces those that do not: a process that read a value back from
# the device BEFORE its first context (r07b's slowdown), with and without the check; and a clean process (the check must cost nothing)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2 3; do for w in 1080p1 vga1 4k1; do for m in clean readback; do for st in 0 1; do
  OATGPU_SETTLE_STREAMS=$st OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_meas.so timeout -k 5 300 python tools/stream_settle_probe.py $m $w 2>&1 | grep " fps "
done; done; done; done
} < /dev/null > $O/r07v_stream_settle.txt 2>&1
cat $O/r07v_stream_settle.txt
