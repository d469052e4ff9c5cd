#!/bin/bash
# tools/lds_phase_probe.py under rocprofv3 --kernel-trace, both modes:   tools/lds_phase_probe.sh OUTDIR [workload]
out=${1:-gpurun_out/ldsp}; wl=${2:-4k1}; R=$PWD; mkdir -p $R/$out
export OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_ldst.so
cd /tmp && export TMPDIR=/tmp
for mode in load alone; do
  rm -rf /tmp/lp_$mode
  timeout 300 rocprofv3 --kernel-trace -d /tmp/lp_$mode -o r -- python $R/tools/lds_phase_probe.py --workload $wl --mode $mode > /tmp/lp_$mode.out 2> /tmp/lp_$mode.err || tail -3 /tmp/lp_$mode.err
  { grep "launches of k_blob_lds" /tmp/lp_$mode.out; python $R/profiles/summarize_rocpd.py $(find /tmp/lp_$mode -name "*.db" | head -1) 260 | grep -E "kernel \||k_blob_lds|k_rowscan|k_mog_fused"; } | tee $R/$out/${wl}_$mode.txt
done
