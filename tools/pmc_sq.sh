#!/bin/bash
# SQ-side PMC counters of k_mog_fused (is it VALU-, wait- or issue-bound?): tools/pmc_sq.sh OUTDIR "bench args"
out=$1; shift
mkdir -p $out; R=$PWD
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/sq_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/sq_$tag -o r -- python $R/bench.py --pmc-child --early-blob 0 "$@" > /dev/null 2> /tmp/sq_$tag.err || tail -3 /tmp/sq_$tag.err
  db=$(find /tmp/sq_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python $R/profiles/summarize_pmc.py $db k_mog_fused 30 >> $R/$out/sq.md
done
cat $R/$out/sq.md
