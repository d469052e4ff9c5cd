#!/bin/bash
# r07x: bench.py pins itself to the GPU's NUMA node BEFORE it first touches the device (the runtime's helper threads inherit the affinity
# of the moment they are created) against after (OATGPU_BENCH_PIN_LATE=1, the order of rounds 2-4)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2 3; do for w in vga1 1080p1 4k1; do for late in 1 0; do
  OATGPU_BENCH_PIN_LATE=$late timeout -k 5 300 python bench.py --workload $w --steps 1000 --quick --check-steps 8 --detail-out $O/r07x_tmp.json > /dev/null 2> $O/r07x_tmp.log < /dev/null
  python - $O/r07x_tmp.json $w $late <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); st = j["stage_ms"]; l = j.get("latency_us") or {}
    print(f"{sys.argv[2]:7s} pin {'after' if sys.argv[3] == '1' else 'BEFORE'} the device is opened: fps {j['value']:9.1f}  step {j['ms_per_step']*1e3:6.2f} us  K1 {st['mog']*1e3:5.1f} us  one frame at a time p50 {l.get('single_p50'):.1f} us  node {j.get('host_numa_node')}")
except Exception as e:
    print(sys.argv[2:], "FAILED", e)
PY
done; done; done
} > $O/r07x_pin_before_open.txt 2>&1
cat $O/r07x_pin_before_open.txt
