#!/bin/bash
# r07w: r07b's slowdown reproduced through bench.py itself (OATGPU_BENCH_OPEN_READBACK=1: the read-back in open_device_with_retry), with the
# library's stream self-check off / on (OATGPU_SETTLE_STREAMS, measurement build)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2; do for w in vga1 1080p1; do for rb in 0 1; do for st in 0 1; do
  OATGPU_BENCH_OPEN_READBACK=$rb OATGPU_SETTLE_STREAMS=$st OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_meas.so timeout -k 5 300 python bench.py --workload $w --steps 1000 --quick --check-steps 8 --detail-out $O/r07w_tmp.json > /dev/null 2> $O/r07w_tmp.log < /dev/null
  python - $O/r07w_tmp.json $w $rb $st <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); st = j["stage_ms"]
    print(f"{sys.argv[2]:7s} readback {sys.argv[3]} settle {sys.argv[4]}: fps {j['value']:9.1f}  step {j['ms_per_step']*1e3:6.2f} us  K1 {st['mog']*1e3:5.1f} us  parity {j['parity']}")
except Exception as e:
    print(sys.argv[2:], "FAILED", e)
PY
done; done; done; done
} > $O/r07w_readback_vs_settle.txt 2>&1
cat $O/r07w_readback_vs_settle.txt
