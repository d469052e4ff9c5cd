#!/bin/bash
# Clocks and power while the dense 4K per-pixel launch runs un-profiled, ONE frame a launch against TWO:
# tools/dense_clocks.sh OUTFILE.  rocm-smi sampled ~3 x a second during 4 000 pipelined steps of each form, interleaved twice.
out=$1; R=$PWD
{
echo "# dense 4K model, pipelined, no profiler: rocm-smi while each form runs (sclk = shader, mclk = HBM, fclk = fabric; socket power)"
for rep in 1 2; do for nf in 1 2; do
  python $R/bench.py --pmc-child --workload 4k1 --dense-model --fusion $nf --steps 60000 --warmup 100 --age 60 > /dev/null 2>&1 &
  pid=$!
  sleep 8
  for i in 1 2 3 4 5 6 7 8; do
    kill -0 $pid 2>/dev/null || break
    c=$(rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk|socclk" | sed -E 's/.*(sclk|mclk|fclk|socclk) clock level: ?[0-9]*:? ?\(([0-9]+Mhz)\).*/\1 \2/' | tr '\n' ' ')
    p=$(rocm-smi --showpower 2>/dev/null | grep -iE "power" | grep -oE "[0-9]+\.[0-9]+" | head -1)
    echo "frames/launch $nf: $c power ${p} W"
    sleep 0.3
  done
  wait $pid
done; done
echo
echo "# the same two forms timed in ONE process, interleaved (tools/dense_placement_probe.py): HIP events around k_mog_fused, sustained state"
OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_meas.so timeout -k 5 300 python $R/tools/dense_placement_probe.py --rounds 1 2>&1 | grep -v amdgpu.ids | cut -c1-100
} > $R/$out 2>&1
cat $R/$out
