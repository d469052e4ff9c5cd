#!/bin/bash
# r07b: (1) where k_rowscan's time beside the per-pixel kernel goes (in-kernel clocks), (2) the small host-bound workloads, this tree
# against round 4's tree (build/r04_tree) on the SAME box, interleaved
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for mode in load alone load; do
  OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_rst.so timeout -k 5 300 python tools/rowscan_probe.py --workload 4k1 --mode $mode 2>&1 | tail -1
done
OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_rst.so timeout -k 5 300 python tools/rowscan_probe.py --workload 1080p1 --mode load 2>&1 | tail -1
} < /dev/null > $O/r07b_rowscan_probe.txt 2>&1
cat $O/r07b_rowscan_probe.txt
{
for rep in 1; do
  for w in vga1 1080p1; do
    ( cd $R/build/r04_tree && timeout -k 5 300 python bench.py --workload $w --steps 1000 --quick --check-steps 8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r04 tree  $w fps %9.1f  ms_per_step %.5f  K1 %.2f us' % (j['value'], j['ms_per_step'], j['stage_ms']['mog']*1e3))" )
    timeout -k 5 300 python bench.py --workload $w --steps 1000 --quick --check-steps 8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('this tree $w fps %9.1f  ms_per_step %.5f  K1 %.2f us' % (j['value'], j['ms_per_step'], j['stage_ms']['mog']*1e3))"
  done
done
} < /dev/null > $O/r07b_small_workloads_vs_r04.txt 2>&1
cat $O/r07b_small_workloads_vs_r04.txt
