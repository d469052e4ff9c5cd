#!/usr/bin/env python3
"""The library's stream self-check (oatgpu_api.hip settle_streams) against a process that used the device BEFORE its first context:

    OATGPU_SETTLE_STREAMS=0|1 OATGPU_MEASURE_PY=1 OATGPU_LIB=build/variants/liboatgpu_meas.so python tools/stream_settle_probe.py clean|readback WORKLOAD

readback: `torch.zeros(8, device=...).sum().item()` first -- what cost one 1080p stream 20 % and a 640 x 480 stream 40 % in r07b.
The self-check itself was measured (profiles/r07v_stream_settle.txt: never anything to replace) and removed: apply
tools/patches/r07_stream_self_check.diff and `make variant NAME=meas` to run this again; without it the last column reads n/a."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    mode, wl = sys.argv[1], sys.argv[2]
    import torch
    torch.cuda.set_device(0)
    if mode == "readback":
        torch.zeros(8, device="cuda").sum().item()
    import bench
    leg = bench.Leg(wl, 0, 0, pool=24)
    K = 1000 if wl in ("vga1", "1080p1") else 400
    tr = bench.timed_run(leg, K, 100, lambda: (leg.hp.synchronize(), torch.cuda.synchronize()), 8, age_frames=300, export=False, spin=0.0)
    k1 = bench.k1_ms(tr["prof"])[0]
    print(f"{mode:9s} settle {os.environ.get('OATGPU_SETTLE_STREAMS', '1')} {wl:8s} fps {leg.ns * K / tr['block_s']:9.1f}  step {tr['block_s'] / K * 1e6:7.2f} us  "
          f"K1 {k1 * 1e3:6.1f} us  streams replaced {getattr(leg.hp.lib, 'oatgpu_streams_replaced', lambda: 'n/a')()}", flush=True)
    leg.close()


if __name__ == "__main__":
    main()
