"""Measurement aid: the bench.py timed loop from a Python process WITHOUT torch (device memory
through ctypes on libamdhip64), to separate the interpreter's cost from torch's.
  OATGPU_NO_TORCH_PRELOAD=1 python tools/loop_notorch.py [--torch] [--threads N]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
use_torch = "--torch" in sys.argv
if use_torch:
    import torch
    if "--threads" in sys.argv:
        torch.set_num_threads(int(sys.argv[sys.argv.index("--threads") + 1]))
    torch.zeros(1, device="cuda")
    if "--cpuop" in sys.argv:
        (torch.rand(4096, 4096) @ torch.rand(4096, 4096)).sum()
else:
    os.environ["OATGPU_NO_TORCH_PRELOAD"] = "1"
from oat_amd import ffi  # noqa: E402

lib = ffi.load()
hip = C.CDLL("libamdhip64.so.7" if use_torch else "/opt/rocm/lib/libamdhip64.so.7")
rows, cols, ns, ring, steps = 1080, 1920, 1, 8, 4000
cfg = ffi.Config()
lib.oatgpu_default_config(C.byref(cfg))
cfg.rows, cfg.cols, cfg.n_streams, cfg.ring_depth = rows, cols, ns, ring
cfg.h_lo, cfg.h_hi, cfg.s_lo, cfg.s_hi, cfg.v_lo, cfg.v_hi = 100, 125, 150, 256, 100, 256
cfg.erode, cfg.dilate, cfg.min_area, cfg.max_area = 3, 7, 20.0, 1e7
ctx = lib.oatgpu_create(C.byref(cfg))
assert ctx, lib.oatgpu_last_error(None)

rng = np.random.default_rng(0)
base = rng.integers(60, 124, (rows, cols, 3), dtype=np.uint8)
ptrs = []
yy, xx = np.mgrid[0:rows, 0:cols]
for f in range(24):
    fr = base + rng.integers(0, 4, base.shape, dtype=np.uint8)
    cx, cy = cols / 2 + cols / 3 * np.cos(0.37 * f), rows / 2 + rows / 3 * np.sin(0.53 * f)
    fr[(xx - cx) ** 2 + (yy - cy) ** 2 <= 60 ** 2] = (220, 120, 30)
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(fr.nbytes)) == 0
    assert hip.hipMemcpy(p, fr.ctypes.data_as(C.c_void_p), C.c_size_t(fr.nbytes), 1) == 0
    ptrs.append(p)

enq, col = lib.oatgpu_track_enqueue_dev, lib.oatgpu_track_collect
buf = (ffi.Position * ns)()


def run(n):
    outstanding = found = 0
    for i in range(n):
        if outstanding == ring:
            assert col(ctx, buf) == 0
            found += buf[0].valid
            outstanding -= 1
        assert enq(ctx, ptrs[i % len(ptrs)], 0.01) == 0
        outstanding += 1
    while outstanding:
        assert col(ctx, buf) == 0
        found += buf[0].valid
        outstanding -= 1
    return found


def run_sequence(n):
    seq = (C.c_void_p * n)(*[ptrs[i % len(ptrs)].value for i in range(n)])
    out = (ffi.Position * (n * ns))()
    t0 = time.perf_counter()
    assert lib.oatgpu_track_sequence_dev(ctx, seq, n, 0.01, out) == 0
    dt = time.perf_counter() - t0
    print(f"  (inside the call: {dt / n * 1e6:.2f} us/step)")
    return sum(out[i].valid for i in range(n * ns))


if "--sequence" in sys.argv:
    run = run_sequence
run(200)
lib.oatgpu_synchronize(ctx)
t0 = time.perf_counter()
found = run(steps)
lib.oatgpu_synchronize(ctx)
dt = time.perf_counter() - t0
print(f"torch={use_torch} argv={sys.argv[1:]} us_per_step={dt / steps * 1e6:.2f} found={found}/{steps}")
