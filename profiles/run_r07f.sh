#!/bin/bash
# r07f: kernel-trace timelines of the 4K pipelined run, row-scan shape 0 (4 x 4) against 4 (8 x 8): where does the step time go?
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
{
for s in 0 4 0 4; do
  rm -rf /tmp/tl_$s
  OATGPU_ROWSCAN_SHAPE=$s OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_meas.so timeout -k 5 300 rocprofv3 --kernel-trace -d /tmp/tl_$s -o r -- python $R/bench.py --pmc-child --workload 4k1 --steps 600 --warmup 100 > /dev/null 2> /tmp/tl_$s.err || tail -3 /tmp/tl_$s.err
  db=$(find /tmp/tl_$s -name "*.db" | head -1)
  echo "--- OATGPU_ROWSCAN_SHAPE=$s"
  python $R/tools/timeline.py $db 400
  python $R/profiles/summarize_rocpd.py $db 400 | grep -E "k_rowscan|k_blob_lds|k_mog_fused|k_publish" | cut -c1-200
done
} < /dev/null > $O/r07f_timeline_rowscan_shapes.txt 2>&1
cat $O/r07f_timeline_rowscan_shapes.txt
