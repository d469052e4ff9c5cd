#!/bin/bash
# r07g: the paired back half (one row-scan launch + one blob launch for both frames of a step): small workloads on / off, GPU tests
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2; do
  for w in vga1 1080p1 1080p2 1080p8 1080p16; do
    for pb in 0 1; do
      OATGPU_PAIR_BACK=$pb OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_meas.so timeout -k 5 300 python bench.py --workload $w --steps 600 --quick --check-steps 8 --detail-out $O/r07g_tmp.json > /dev/null 2> $O/r07g_tmp.log < /dev/null
      python - $O/r07g_tmp.json $w $pb <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    st = j["stage_ms"]; l = j.get("latency_us") or {}
    print(f"{sys.argv[2]:8s} paired {sys.argv[3]}: fps {j['value']:9.1f}  step {j['ms_per_step']*1e3:7.2f} us  K1 {st['mog']*1e3:6.1f} us  back half {st['blob']*1e3:6.1f} us  gpu_total {st['gpu_total']*1e3:6.1f} us  single p50 {l.get('single_p50'):.1f}  parity {j['parity']}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
    done
  done
done
} > $O/r07g_paired_back_half_ab.txt 2>&1
cat $O/r07g_paired_back_half_ab.txt
( timeout -k 5 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | cut -c1-400 ) < /dev/null > $O/r07g_gputests.txt 2>&1
cat $O/r07g_gputests.txt
