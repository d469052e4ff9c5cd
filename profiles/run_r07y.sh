#!/bin/bash
# r07y: rows a row-scan workgroup takes (four waves; 4 = shipped, 8, 16): fewer workgroups to dispatch beside K1, a longer loop in each
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2; do for w in 4k1 1080p1 vga1 1080p8; do for v in meas rr8 rr16; do
  OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_$v.so timeout -k 5 300 python bench.py --workload $w --steps 600 --quick --check-steps 8 --detail-out $O/r07y_tmp.json > /dev/null 2> $O/r07y_tmp.log < /dev/null
  python - $O/r07y_tmp.json $w $v <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); st = j["stage_ms"]; l = j.get("latency_us") or {}
    print(f"{sys.argv[2]:7s} {sys.argv[3]:5s}: fps {j['value']:9.1f}  K1 {st['mog']*1e3:6.1f}  row scan + blob {st['blob']*1e3:6.1f}  gpu_total {st['gpu_total']*1e3:6.1f} us  one frame at a time p50 {l.get('single_p50'):.1f}  saturated p50 {l.get('saturated_p50'):.1f}  parity {j['parity']}")
except Exception as e:
    print(sys.argv[2:], "FAILED", e)
PY
done; done; done
} > $O/r07y_rowscan_rows_per_workgroup_ab.txt 2>&1
cat $O/r07y_rowscan_rows_per_workgroup_ab.txt
cd /tmp; export TMPDIR=/tmp
{
for v in meas rr8 rr16; do
  rm -rf /tmp/tl_r
  OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_$v.so timeout -k 5 300 rocprofv3 --kernel-trace -d /tmp/tl_r -o r -- python $R/bench.py --pmc-child --workload 4k1 --steps 600 --warmup 100 > /dev/null 2> /tmp/tl_r.err || tail -3 /tmp/tl_r.err
  db=$(find /tmp/tl_r -name "*.db" | head -1)
  echo "--- $v"
  python $R/tools/timeline.py $db 400
done
} < /dev/null > $O/r07y_timeline.txt 2>&1
cat $O/r07y_timeline.txt
