#!/bin/bash
# r07r: does RCCL initialised BEFORE the library's streams exist (what an N > 1 bench.py rank did) cost the hot path?  One GPU, world of one
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2; do for w in vga1 1080p1 4k1 1080p8; do for m in none nccl_first context_first; do
  timeout -k 5 300 python tools/nccl_order_probe.py $m $w 2>&1 | grep " fps " 
done; done; done
} < /dev/null > $O/r07r_rccl_init_order.txt 2>&1
cat $O/r07r_rccl_init_order.txt
