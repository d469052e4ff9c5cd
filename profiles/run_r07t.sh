#!/bin/bash
# r07t: a frame launched with nothing else outstanding takes the plain order (OATGPU_LONE_PLAIN=1) against the early order (0): one-frame-at-a-time latency, frame rate
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2; do for w in 4k1 1080p2; do for lp in 0 1; do
  OATGPU_LONE_PLAIN=$lp OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_meas.so timeout -k 5 300 python bench.py --workload $w --steps 600 --quick --check-steps 8 --detail-out $O/r07t_tmp.json > /dev/null 2> $O/r07t_tmp.log < /dev/null
  python - $O/r07t_tmp.json $w $lp <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); st = j["stage_ms"]; l = j.get("latency_us") or {}
    print(f"{sys.argv[2]:7s} lone-plain {sys.argv[3]}: fps {j['value']:9.1f}  one frame at a time p50 {l.get('single_p50'):.1f} p99 {l.get('single_p99'):.1f} us  saturated p50 {l.get('saturated_p50'):.1f}  parity {j['parity']}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
done; done; done
} > $O/r07t_lone_frame_plain_order_ab.txt 2>&1
cat $O/r07t_lone_frame_plain_order_ab.txt
