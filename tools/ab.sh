#!/bin/bash
# A/B of kernel variants on ONE box: tools/ab.sh OUTDIR "variant names ('' = product lib)" "bench arg sets separated by ;"
# e.g. tools/ab.sh gpurun_out/ab1 "default planar t256" "--workload 4k1 --dense-model --steps 300;--workload 4k1 --steps 1000"
out=$1; variants=$2; IFS=';' read -ra sets <<< "$3"
mkdir -p "$out"
for v in $variants; do
  for i in "${!sets[@]}"; do
    lib=build/variants/liboatgpu_$v.so; [ "$v" = default ] && lib=oat_amd/lib/liboatgpu.so
    OATGPU_MEASURE_PY=1 OATGPU_LIB=$PWD/$lib python bench.py ${sets[$i]} --quick --check-steps 16 $AB_EXTRA --detail-out "$out/${v}_$i.json" > "$out/${v}_$i.line" 2> "$out/${v}_$i.log"
    python - "$out/${v}_$i.json" "$v" "${sets[$i]}" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    r = j["roofline"]
    print(f"{sys.argv[2]:12s} {sys.argv[3]:48s} fps {j['value']:9.1f}  step {j['ms_per_step']*1e3:7.1f} us  K1 {r['benched_workload']['avg_launch_ms']*1e3:7.1f} us  blob {j['stage_ms']['blob']*1e3:6.1f} us  parity {j['parity']}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  done
done
