#!/bin/bash
# r07d: the small host-bound workloads after the device-open fix, this tree against round 4's tree, same box, interleaved
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2; do
  for w in vga1 1080p1; do
    ( cd $R/build/r04_tree && timeout -k 5 300 python bench.py --workload $w --steps 1000 --quick --check-steps 8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r04 tree  $w fps %9.1f  ms_per_step %.5f  K1 %.2f us' % (j['value'], j['ms_per_step'], j['stage_ms']['mog']*1e3))" )
    timeout -k 5 300 python bench.py --workload $w --steps 1000 --quick --check-steps 8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('this tree $w fps %9.1f  ms_per_step %.5f  K1 %.2f us' % (j['value'], j['ms_per_step'], j['stage_ms']['mog']*1e3))"
  done
done
} < /dev/null > $O/r07d_small_workloads_vs_r04.txt 2>&1
cat $O/r07d_small_workloads_vs_r04.txt
