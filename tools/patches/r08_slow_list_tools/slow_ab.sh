#!/bin/bash
# The slow list on / off with ONE binary on one box: tools/slow_ab.sh OUT [reps] [tests: 1|0]
out=${1:-gpurun_out/slowab}; reps=${2:-2}; tests=${3:-1}
mkdir -p $out
if [ "$tests" = 1 ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_zz_long_run_gpu.py -x -q -m gpu -k "two_frames or full_size or determinism or long_run or hot_path or single_launch or fusion or mog2_mask or slow" > $out/tests.txt 2>&1; tail -5 $out/tests.txt
fi
for r in $(seq $reps); do
  for sl in 0 1; do
    for set in "--workload 4k1 --steps 1000" "--workload 1080p16 --steps 200 --warmup 40" "--workload 1080p1 --steps 1500" "--workload 4k1 --steps 600 --fusion 1" "--workload 4k1 --steps 1000 --learning-rate 0"; do
      python bench.py $set --quick --check-steps 16 --slow-list $sl --detail-out "$out/s${sl}_$r.json" > "$out/s${sl}_$r.line" 2> "$out/s${sl}_$r.log"
      python - "$out/s${sl}_$r.json" "slow=$sl" "$set" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    r = j["roofline"]
    print(f"{sys.argv[2]:8s} {sys.argv[3]:48s} fps {j['value']:9.1f}  step {j['ms_per_step']*1e3:7.1f} us  K1 {r['benched_workload']['avg_launch_ms']*1e3:7.1f} us  blob {j['stage_ms']['blob']*1e3:6.1f} us  parity {j['parity']}  listed {j.get('slow_list_entries')}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
    done
  done
done | tee $out/ab.txt
