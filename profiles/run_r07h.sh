#!/bin/bash
# r07h: the early order with FOUR scratch sets (a step's row scans wait for nothing but their own per-pixel launch) against two
# (r04-r06), crossed with the row-scan workgroup shape; timelines out of kernel traces; then the GPU tests
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
cd $R
{
for rep in 1 2; do
for sets in 2 4; do for s in 0 4; do
  OATGPU_EARLY_SETS=$sets OATGPU_ROWSCAN_SHAPE=$s OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_meas.so timeout -k 5 300 python bench.py --workload 4k1 --steps 1000 --quick --check-steps 16 --detail-out $O/r07h_tmp.json > /dev/null 2> $O/r07h_tmp.log < /dev/null
  python - $O/r07h_tmp.json $sets $s <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    st = j["stage_ms"]; l = j.get("latency_us") or {}
    print(f"sets {sys.argv[2]} rowscan shape {sys.argv[3]}: fps {j['value']:9.1f}  K1 {st['mog']*1e3:6.1f} us  rowscan+blob {st['blob']*1e3:6.1f} us  gpu_total {st['gpu_total']*1e3:6.1f} us  single p50 {l.get('single_p50'):.1f}  saturated p50 {l.get('saturated_p50'):.1f}  parity {j['parity']}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
done; done; done
} > $O/r07h_early_sets_ab.txt 2>&1
cat $O/r07h_early_sets_ab.txt
cd /tmp; export TMPDIR=/tmp
{
for sets in 2 4; do for s in 0 4; do
  rm -rf /tmp/tl_$s
  OATGPU_EARLY_SETS=$sets OATGPU_ROWSCAN_SHAPE=$s OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_meas.so timeout -k 5 300 rocprofv3 --kernel-trace -d /tmp/tl_$s -o r -- python $R/bench.py --pmc-child --workload 4k1 --steps 600 --warmup 100 > /dev/null 2> /tmp/tl_$s.err || tail -3 /tmp/tl_$s.err
  db=$(find /tmp/tl_$s -name "*.db" | head -1)
  echo "--- OATGPU_EARLY_SETS=$sets OATGPU_ROWSCAN_SHAPE=$s"
  python $R/tools/timeline.py $db 400
done; done
} < /dev/null > $O/r07h_timeline.txt 2>&1
cat $O/r07h_timeline.txt
cd $R
( timeout -k 5 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | cut -c1-400 ) < /dev/null > $O/r07h_gputests.txt 2>&1
cat $O/r07h_gputests.txt
