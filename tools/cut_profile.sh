#!/bin/bash
# Dynamic vector / scalar instructions a wave of k_mog_fused by region (tools/cut_profile.py): tools/cut_profile.sh OUT [--fusion1]
# needs build/variants/liboatgpu_cut{1..5}.so and liboatgpu_meas.so (make variant NAME=cut$n DEFS=-DOATGPU_CUT=$n; make variant NAME=meas)
out=${1:-gpurun_out/cut}; shift
R=$PWD; mkdir -p $R/$out
python tools/cut_profile.py save $R/$out/model > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for v in cut1 cut2 cut3 cut4 cut5 meas; do
  rm -rf /tmp/cp_$v
  OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_$v.so timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/cp_$v -o r -- python $R/tools/cut_profile.py run "$@" $R/$out/model > /dev/null 2> /tmp/cp_$v.err || tail -3 /tmp/cp_$v.err
  db=$(find /tmp/cp_$v -name "*.db" | head -1)
  echo "## $v" >> $R/$out/cut.md
  [ -n "$db" ] && python $R/profiles/summarize_pmc.py $db k_mog_fused 8 >> $R/$out/cut.md
done
cat $R/$out/cut.md
