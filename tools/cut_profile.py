#!/usr/bin/env python3
"""Dynamic instruction profile of the per-pixel kernel by truncation (measurement aid, LABNOTES r03).

    python tools/cut_profile.py save  [--dense] DIR      product library: age a 4K model as bench.py does, save it + 8 frames
    python tools/cut_profile.py run   [--dense] [--fusion1] DIR      (OATGPU_LIB = a -DOATGPU_CUT=n variant, under rocprofv3 --pmc
                                                          SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES): load, 24 two-frame launches

A -DOATGPU_CUT=n build ends k_mog_fused at cut n with everything computed so far kept alive through one store, and
stores nothing else: every launch sees the same aged model.  SQ_INSTS_VALU / SQ_WAVES of the variants, differenced,
is the dynamic vector instruction count of each region of the kernel on that model."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench


def main():
    mode, dense, d = sys.argv[1], "--dense" in sys.argv, sys.argv[-1]
    if "--quiet" in sys.argv:                      # the dense leg whose pixels all stay background (bench.py roofline.all_background)
        bench.DENSE_NOISE = 3
    os.makedirs(d, exist_ok=True)
    tag = ("quiet" if "--quiet" in sys.argv else "dense") if dense else "sparse"
    if mode == "save":
        leg = bench.Leg("4k1", 0, 0, dense=dense, pool=10 if dense else 48)
        leg.init()
        leg.age(120 if dense else 600)
        leg.hp.save_mog_state(os.path.join(d, f"{tag}.mog"))
        fr = [leg.pool[leg.pool_index(leg.step + i)].cpu().numpy() for i in range(8)]
        np.save(os.path.join(d, f"{tag}_frames.npy"), np.stack(fr))
        leg.close()
        return
    hp = bench.make_hotpath(bench.WORKLOADS["4k1"], 0, dense=dense)
    hp.load_mog_state(os.path.join(d, f"{tag}.mog"))
    hp.set_fusion(1 if "--fusion1" in sys.argv else 2)     # --fusion1: the one-frame-a-launch instantiation
    hp.set_early_blob(False)                               # rocprofv3 --pmc serialises dispatches (oatgpu_set_early_blob)
    fr = [torch.from_numpy(f).cuda() for f in np.load(os.path.join(d, f"{tag}_frames.npy"))]
    torch.cuda.synchronize()
    if dense:                      # the library picks the streaming-load instantiation from its density probes (frames 8, 16, ..)
        pass
    for i in range(48):
        hp.enqueue_dev(fr[i % 8].data_ptr())
        if hp.outstanding() == 4:
            hp.collect()
    while hp.outstanding():
        hp.collect()
    hp.close()


if __name__ == "__main__":
    main()
