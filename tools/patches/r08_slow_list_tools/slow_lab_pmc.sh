#!/bin/bash
# SQ instruction counters of the slowlab9 variant under different lab bits: tools/slow_lab_pmc.sh OUT "bits ..."
out=${1:-gpurun_out/r08g}; bitsets=${2:-"0x1000 0x1900"}
mkdir -p $out; R=$PWD
cd /tmp && export TMPDIR=/tmp
for b in $bitsets; do
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM"; do
    tag=$(echo $set | cut -d' ' -f1)_$b
    rm -rf /tmp/sq_$tag
    OATGPU_SLOWLAB_BITS=$b OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_slowlab9.so timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/sq_$tag -o r -- python $R/bench.py --pmc-child --early-blob 0 --workload 4k1 --steps 100 --warmup 20 --age 600 > /dev/null 2> /tmp/sq_$tag.err || tail -3 /tmp/sq_$tag.err
    db=$(find /tmp/sq_$tag -name "*.db" | head -1)
    echo "## bits $b" >> $R/$out/sq.md
    [ -n "$db" ] && python $R/profiles/summarize_pmc.py $db k_mog_fused 30 >> $R/$out/sq.md
  done
done
cat $R/$out/sq.md
