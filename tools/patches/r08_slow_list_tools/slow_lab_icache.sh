#!/bin/bash
# instruction-cache counters of k_mog_fused with the slow list on / off (the shipped library)
out=${1:-gpurun_out/r08h}
mkdir -p $out; R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_DCACHE[A-Z_]*\|SQ_INSTS_SMEM\|SQ_WAIT_IFETCH\|SQ_INST_LEVEL_[A-Z]*" | sort -u | tr '\n' ' ' > $R/$out/avail.txt
cat $R/$out/avail.txt; echo
for sl in 0 1; do
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SMEM" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
    tag=$(echo $set | cut -d' ' -f1)_$sl
    rm -rf /tmp/sq_$tag
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/sq_$tag -o r -- python $R/bench.py --pmc-child --early-blob 0 --workload 4k1 --steps 100 --warmup 20 --age 600 --slow-list $sl > /dev/null 2> /tmp/sq_$tag.err || tail -3 /tmp/sq_$tag.err
    db=$(find /tmp/sq_$tag -name "*.db" | head -1)
    echo "## slow list $sl" >> $R/$out/sq.md
    [ -n "$db" ] && python $R/profiles/summarize_pmc.py $db k_mog_fused 30 >> $R/$out/sq.md
  done
done
cat $R/$out/sq.md
