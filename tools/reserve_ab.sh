#!/bin/bash
# Stream A kept off n compute units (OATGPU_A_RESERVE, measurement build liboatgpu_meas.so):  tools/reserve_ab.sh OUT "0 4 8 16"
out=${1:-gpurun_out/rsv}; ns=${2:-"0 4 8"}; R=$PWD; mkdir -p $out
export OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_meas.so
for n in $ns; do
  export OATGPU_A_RESERVE=$n
  for set in "--workload 4k1 --steps 1000" "--workload 1080p16 --steps 200 --warmup 40" "--workload 1080p1 --steps 1500"; do
    python bench.py $set --quick --check-steps 16 --detail-out $out/r${n}.json > $out/r${n}.line 2> $out/r${n}.log
    python - $out/r${n}.json "reserve $n" "$set" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); r = j["roofline"]
    print(f"{sys.argv[2]:12s} {sys.argv[3]:48s} fps {j['value']:9.1f}  step {j['ms_per_step']*1e3:7.1f} us  K1 {r['benched_workload']['avg_launch_ms']*1e3:7.1f} us  blob {j['stage_ms']['blob']*1e3:6.1f} us  gpu_total {j['stage_ms'].get('gpu_total', 0)*1e3:6.1f} us  parity {j['parity']}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  done
  echo "== ktrace 4k1 pipelined, reserve $n"
  bash tools/ktrace.sh $out/kt_r$n.md --workload 4k1 --steps 1000 --warmup 40 | grep -E "k_blob_lds|k_mog_fused|k_rowscan"
done 2>&1 | tee $out/ab.txt
