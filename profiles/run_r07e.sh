#!/bin/bash
# r07e: row-scan workgroup shapes with MORE waves a workgroup (fewer workgroups to dispatch beside the per-pixel kernel)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for s in 0 4 5 6 0 4; do
  echo "--- OATGPU_ROWSCAN_SHAPE=$s  (0: 4 waves x 4 rows, 4: 8 x 8, 5: 16 x 16, 6: 2 x 2)"
  OATGPU_ROWSCAN_SHAPE=$s OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_rst.so timeout -k 5 300 python tools/rowscan_probe.py --workload 4k1 --mode load --steps 200 2>&1 | tail -1
  OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_rst.so OATGPU_ROWSCAN_SHAPE=$s timeout -k 5 300 python bench.py --workload 4k1 --steps 1000 --quick --check-steps 16 --detail-out $O/r07e_rs$s.json > /dev/null 2> $O/r07e_rs$s.log < /dev/null
  python - $O/r07e_rs$s.json $s <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    st = j["stage_ms"]; l = j.get("latency_us") or {}
    print(f"bench shape {sys.argv[2]}: fps {j['value']:9.1f} K1 {st['mog']*1e3:6.1f} us  rowscan+blob {st['blob']*1e3:6.1f} us  gpu_total {st['gpu_total']*1e3:6.1f} us  single p50 {l.get('single_p50'):.1f} sat p50 {l.get('saturated_p50'):.1f} parity {j['parity']}")
except Exception as e:
    print("shape", sys.argv[2], "FAILED", e)
PY
done
} > $O/r07e_rowscan_shape_ab.txt 2>&1
cat $O/r07e_rowscan_shape_ab.txt
