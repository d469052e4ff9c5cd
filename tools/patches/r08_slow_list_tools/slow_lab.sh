#!/bin/bash
# lab A/B of the slow list on one box (results of the slowlab variants are invalid by construction: timing only)
# tools/slow_lab.sh OUT "bits bits ..."  -- OATGPU_SLOWLAB_BITS values for the slowlab9 variant
out=${1:-gpurun_out/r08f}; bitsets=${2:-"0x1000 0x1100 0x1300 0x1500 0x1900 0x1f00"}
mkdir -p $out
for r in 1 2; do
AB_EXTRA="--no-parity" bash tools/ab.sh $out/base$r "noslow" "--workload 4k1 --steps 1000"
for b in $bitsets; do
echo "bits $b"
OATGPU_SLOWLAB_BITS=$b AB_EXTRA="--no-parity" bash tools/ab.sh $out/b${b}_$r "slowlab9" "--workload 4k1 --steps 1000"
done
done 2>&1 | tee $out/ab.txt
