for v in ${VARIANTS:-default default}; do
  lib=build/variants/liboatgpu_$v.so; [ "$v" = default ] && lib=oat_amd/lib/liboatgpu.so
  OATGPU_MEASURE_PY=1 OATGPU_LIB=$PWD/$lib python bench.py --workload 4k1 --steps 300 --quick --check-steps 16 --detail-out /tmp/det_check.json > /dev/null 2>&1; cat /tmp/det_check.json | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']['benched_workload']
print('$v', r['useful_bytes_per_px'], r['mode_histogram']['live_modes'], j['parity'])"
done
