#!/bin/bash
# Why does the dense ONE-frame launch of the per-pixel kernel lose to its two-frame sibling?  
# Same box, interleaved: counters of k_mog_fused<3,0,1,NF,0,64> for NF = 1 and 2 on the dense 4K model -- SQ issue / wait,
# vector-memory instruction counts, the L2's (TCC) requests to and stalls on the memory side, the L1's (TCP) pending stalls --
# and the memory / shader clocks sampled while each runs without a profiler.       tools/dense_one_vs_two.sh OUTFILE
out=$1; R=$PWD; mkdir -p $(dirname $R/$out)
cd /tmp && export TMPDIR=/tmp
avail=$(rocprofv3 -L 2>/dev/null | tr ',: \t' '\n\n\n\n' | sort -u)
pick() { for c in "$@"; do echo "$avail" | grep -qx "$c" && printf "%s " $c; done; }
sets=(
 "$(pick SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR)"
 "$(pick SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR)"
 "$(pick TCC_EA_WRREQ_STALL_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum)"
 "$(pick TCC_EA_WRREQ_IO_CREDIT_STALL_sum TCC_EA_WRREQ_GMI_CREDIT_STALL_sum TCC_EA_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA_RDREQ_DRAM_CREDIT_STALL_sum)"
 "$(pick TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum)"
 "$(pick TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_WRITEBACK_sum)"
 "$(pick GRBM_GUI_ACTIVE GRBM_COUNT)"
)
{
echo "# dense 4K model, k_mog_fused one wave a workgroup: ONE frame a launch (--fusion 1) against TWO (--fusion 2), same box, interleaved"
for rep in 1 2; do for nf in 1 2; do
  echo; echo "## pass $rep, $nf frame(s) a launch"
  for set in "${sets[@]}"; do
    [ -z "$set" ] && continue
    rm -rf /tmp/dv
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/dv -o r -- python $R/bench.py --pmc-child --workload 4k1 --dense-model --early-blob 0 --k1-wg 64 --fusion $nf --steps 60 --warmup 60 --age 60 > /dev/null 2> /tmp/dv.err || tail -3 /tmp/dv.err
    db=$(find /tmp/dv -name "*.db" | head -1)
    [ -n "$db" ] && python $R/profiles/summarize_pmc.py $db k_mog_fused 70 | grep -v "^| kernel\|^|---" | sed 's/void oatgpu:://'
  done
done; done
echo; echo "## clocks while each form runs WITHOUT a profiler (rocm-smi --showclocks every 0.3 s; 3 000 steps)"
for rep in 1 2; do for nf in 1 2; do
  python $R/bench.py --pmc-child --workload 4k1 --dense-model --k1-wg 64 --fusion $nf --steps 3000 --warmup 100 --age 60 > /dev/null 2>&1 &
  pid=$!
  sleep 6
  for i in 1 2 3 4 5 6; do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | tr -s ' ' | cut -d' ' -f3- | tr '\n' ';'; echo; sleep 0.3; kill -0 $pid 2>/dev/null || break; done | sed "s/^/fusion $nf: /"
  wait $pid
done; done
} > $R/$out 2>&1
cat $R/$out
