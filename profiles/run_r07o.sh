#!/bin/bash
# r07o: cache-policy classes of the per-pixel kernel (tools/patches/r07_cache_policy_classes.diff): the scope bit sc1 (aux 16) on the
# model's loads / stores.  Dense one- and two-frame launches (p0-p5) and the everyday 4K workload (p0, p6-p9); fresh process each, interleaved
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
echo "# p0 shipped policies | p1 dense-1: mode 0 loads+stores sc1 | p2 dense-1: mode 0 stores sc1 | p3 = p1 on both dense forms | p4 = p3 + slots 1-4 nt+sc1 | p5 = p3 + slots 1-4 sc1"
for rep in 1 2 3; do for v in p0 p1 p2 p3 p4 p5; do
  OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_$v.so timeout -k 5 200 python tools/dense_placement_probe.py --rounds 1 --forms 1:1:64,2:1:64 2>&1 | grep "k_mog_fused" | cut -c1-85 | sed "s/^/$v  /"
done; done
echo
echo "# everyday 4K workload (bench.py --workload 4k1 --quick): p0 shipped (mode 0 nt, slots 1-4 default loads / nt stores) | p6 mode 0 sc1 | p7 mode 0 nt+sc1 | p8 = p6 + slot 1-4 stores sc1 | p9 = p8 + slot 1-4 loads sc1"
for rep in 1 2; do for v in p0 p6 p7 p8 p9; do
  OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_$v.so timeout -k 5 300 python bench.py --workload 4k1 --steps 1000 --quick --check-steps 16 --detail-out $O/r07o_tmp.json > /dev/null 2> $O/r07o_tmp.log < /dev/null
  python - $O/r07o_tmp.json $v <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); st = j["stage_ms"]
    print(f"{sys.argv[2]}: fps {j['value']:9.1f}  K1 {st['mog']*1e3:6.1f} us  parity {j['parity']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done; done
} > $O/r07o_cache_policy_classes_ab.txt 2>&1
cat $O/r07o_cache_policy_classes_ab.txt
