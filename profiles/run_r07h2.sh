#!/bin/bash
# r07h2: PCIe-inclusive rates (bench.py --input host), this tree against round 4's (build/r04_tree), same box, interleaved
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2; do for w in 1080p1 4k1; do
  ( cd $R/build/r04_tree && timeout 300 python bench.py --workload $w --input host --steps 300 --quick --no-parity 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r04 tree  $w --input host: %.0f fps, %.1f us a step' % (j['value'], j['ms_per_step']*1e3))" )
  timeout 300 python bench.py --workload $w --input host --steps 300 --quick --no-parity --detail-out $O/h2_tmp.json > /dev/null 2>&1; python -c "
import json; j=json.load(open('$O/h2_tmp.json')); print('this tree $w --input host: %.0f fps, %.1f us a step' % (j['value'], j['ms_per_step']*1e3))"
done; done
} 2>&1 | tee $O/r07_host_input_vs_r04.txt
