#!/bin/bash
# r07a (round 5, first GPU session): smoke, row-scan workgroup shape A/B beside the one-wave per-pixel kernel, GPU tests, default bench line
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
( timeout -k 5 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) < /dev/null > $O/r07a_smoke.txt
cat $O/r07a_smoke.txt
{
for s in 0 1 2 3 0 2; do
  OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_meas.so OATGPU_ROWSCAN_SHAPE=$s timeout -k 5 300 python bench.py --workload 4k1 --steps 1000 --quick --check-steps 16 --detail-out $O/r07a_rs$s.json > $O/r07a_rs$s.line 2> $O/r07a_rs$s.log < /dev/null
  python - $O/r07a_rs$s.json $s <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    st = j["stage_ms"]; l = j.get("latency_us") or {}
    print(f"rowscan shape {sys.argv[2]}: fps {j['value']:9.1f} K1 {st['mog']*1e3:6.1f} us  rowscan+blob {st['blob']*1e3:6.1f} us  gpu_total {st['gpu_total']*1e3:6.1f} us  single p50 {l.get('single_p50')} sat p50 {l.get('saturated_p50')} parity {j['parity']}")
except Exception as e:
    print("shape", sys.argv[2], "FAILED", e)
PY
done
} > $O/r07a_rowscan_shape_ab.txt 2>&1
cat $O/r07a_rowscan_shape_ab.txt
( timeout -k 5 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | cut -c1-400 ) < /dev/null > $O/r07a_gputests.txt 2>&1
cat $O/r07a_gputests.txt
timeout -k 5 900 python bench.py < /dev/null > $O/r07a_bench_default.line 2> $O/r07a_bench_default.log
cp bench_detail.json $O/r07a_bench_default_detail.json
wc -c $O/r07a_bench_default.line; cat $O/r07a_bench_default.line
