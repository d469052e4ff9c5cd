#!/bin/bash
# bench lines WITH the PMC child passes for several library variants on one box: tools/pmc_ab.sh OUTDIR "variants"
out=$1; mkdir -p $out
for v in $2; do
  lib=build/variants/liboatgpu_$v.so; [ "$v" = default ] && lib=oat_amd/lib/liboatgpu.so
  OATGPU_MEASURE_PY=1 OATGPU_LIB=$PWD/$lib python bench.py --no-extra --no-cpu-baseline --check-steps 16 --detail-out $out/$v.json > $out/$v.line 2> $out/$v.log
  python - $out/$v.json $v <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); r = j["roofline"]; b = r["benched_workload"]
print(f"{sys.argv[2]:10s} fps {j['value']:8.1f} K1 {b['avg_launch_ms']*1e3:6.1f} us moved {b.get('moved_bytes_per_px',0):5.1f} B/px  dense sustained {r['avg_launch_ms']*1e3:6.1f} us burst {r.get('avg_launch_ms_burst',0)*1e3:6.1f} us dense traffic {r['traffic']/1e6:7.1f} MB parity {j['parity']}")
PY
done
