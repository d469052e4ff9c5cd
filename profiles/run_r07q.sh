#!/bin/bash
# r07q: PARKED row scan in the early order (OATGPU_RS_PARK = persistent row-scan workgroups a stream, dispatched ahead of their
# per-pixel launch, released by a stream memory operation behind it) against the shipped order; bench stage times + timelines
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2; do for park in 0 135 270 68; do
  OATGPU_RS_PARK=$park OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_meas.so timeout -k 5 300 python bench.py --workload 4k1 --steps 1000 --quick --check-steps 16 --detail-out $O/r07q_tmp.json > /dev/null 2> $O/r07q_tmp.log < /dev/null
  python - $O/r07q_tmp.json $park <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    st = j["stage_ms"]; l = j.get("latency_us") or {}
    print(f"park {sys.argv[2]:>3}: fps {j['value']:9.1f}  K1 {st['mog']*1e3:6.1f} us  gpu_total {st['gpu_total']*1e3:6.1f} us  single p50 {l.get('single_p50'):.1f}  saturated p50 {l.get('saturated_p50'):.1f}  timeouts {j.get('early_blob_timeouts')}  parity {j['parity']}")
except Exception as e:
    print("park", sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json', '.log')).read()[-800:])
PY
done; done
} > $O/r07q_parked_rowscan_ab.txt 2>&1
cat $O/r07q_parked_rowscan_ab.txt
cd /tmp; export TMPDIR=/tmp
{
for park in 0 135 270; do
  rm -rf /tmp/tl_p
  OATGPU_RS_PARK=$park OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_meas.so timeout -k 5 300 rocprofv3 --kernel-trace -d /tmp/tl_p -o r -- python $R/bench.py --pmc-child --workload 4k1 --steps 600 --warmup 100 > /dev/null 2> /tmp/tl_p.err || tail -3 /tmp/tl_p.err
  db=$(find /tmp/tl_p -name "*.db" | head -1)
  echo "--- OATGPU_RS_PARK=$park"
  python $R/tools/timeline.py $db 400
done
} < /dev/null > $O/r07q_timeline.txt 2>&1
cat $O/r07q_timeline.txt
