#!/bin/bash
# One-frame-a-launch instantiations of K1 (r04): tools/k1_nf1_ab.sh OUT "variant[:lds_bytes] ..." [reps] [legs]
#   variant = default | a `make variant` name; lds_bytes = OATGPU_K1_LDS (unused dynamic LDS that holds the occupancy down)
#   legs: d = dense 4K model, s = everyday 4K model, z = everyday at learning rate 0 (Z: two frames a launch)
out=${1:-gpurun_out/nf1}; combos=${2:-"default"}; reps=${3:-2}; legs=${4:-"d s"}
mkdir -p $out
for r in $(seq $reps); do
  for c in $combos; do
    v=${c%%:*}; lds=""; [[ $c == *:* ]] && lds=${c##*:}
    lib=build/variants/liboatgpu_$v.so; [ "$v" = default ] && lib=oat_amd/lib/liboatgpu.so
    for leg in $legs; do
      case $leg in
        d) args="--workload 4k1 --dense-model --fusion 1 --steps 300 --warmup 1200 --quick --no-parity";;
        s) args="--workload 4k1 --fusion 1 --steps 1000 --quick --check-steps 16";;
        z) args="--workload 4k1 --fusion 1 --steps 1000 --quick --check-steps 16 --learning-rate 0 --age 60";;
        Z) args="--workload 4k1 --fusion 2 --steps 1000 --quick --check-steps 16 --learning-rate 0 --age 60";;
        D) args="--workload 4k1 --dense-model --fusion 2 --steps 300 --warmup 1200 --quick --no-parity";;
        S) args="--workload 4k1 --fusion 2 --steps 1000 --quick --check-steps 16";;
      esac
      f=$out/${v}_${lds:-0}_${leg}_$r
      OATGPU_K1_LDS=$lds OATGPU_MEASURE_PY=1 OATGPU_LIB=$PWD/$lib python bench.py $args --detail-out $f.json > $f.line 2> $f.log
      python - $f.json "$c" $leg <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    b = j["roofline"]["benched_workload"]
    print(f"{sys.argv[2]:18s} {sys.argv[3]} fps {j['value']:9.1f}  step {j['ms_per_step']*1e3:7.1f} us  K1 {b['avg_launch_ms']*1e3:7.1f} us (fpl {b['frames_per_launch']:.2f})  parity {j['parity']}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
    done
  done
done | tee -a $out/ab.txt
