#!/bin/bash
# r07m: cache policies of the dense ONE-frame per-pixel launch (tools/patches/r07_dense_one_frame_cache_policy.diff), interleaved, fresh process each
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
echo "# dense 4K model, ONE frame a launch (k_mog_fused<3,0,1,1,0,64>), HIP events around the kernel, sustained state; two-frame sibling beside it"
echo "# meas = shipped policies (mode 0 default policy, slots 1-4 streaming loads + stores); d1a = mode 0 loads + stores streaming; d1b = mode 0 stores streaming;"
echo "# d1e = slots 1-4 stores nt+sc1 (aux 18); d1f = nt+sc0 (3); d1g = nt+sc0+sc1 (19); d1h = every store nt+sc1; d1i = mode 0 loads + stores sc1 (16)"
for rep in 1 2; do for v in meas d1e d1f d1g d1h d1i; do
  OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_$v.so timeout -k 5 200 python tools/dense_placement_probe.py --rounds 1 --forms 1:1:64,2:1:64 2>&1 | grep "k_mog_fused" | cut -c1-85 | sed "s/^/$v  /"
done; done
} > $O/r07n_dense_one_frame_store_policy_ab.txt 2>&1
cat $O/r07n_dense_one_frame_store_policy_ab.txt
