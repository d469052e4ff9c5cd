#!/usr/bin/env python3
"""Wall-clock throughput of the drop-in PROCESS pipeline over shared memory (the reference's own
perf methodology, test/perf/*.sh: `time` a free-running frame server with the consumer chain attached).

    python tools/pipeline_fps.py [--rows 480 --cols 640] [--frames 1000] [--fused]
"""
import argparse
import os
import subprocess
import sys
import time
import uuid

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "build", "bin")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--cols", type=int, default=640)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--fused", action="store_true")
    a = ap.parse_args()
    from oat_amd.synth import SyntheticStream
    st = SyntheticStream(a.rows, a.cols, 0, n_discs=2)
    pool = np.stack([st.frame(t, with_discs=t > 0) for t in range(16)])
    raw = f"/dev/shm/oat_pipe_{uuid.uuid4().hex[:8]}.raw"
    pool.tofile(raw)
    tag = "oat_p_" + uuid.uuid4().hex[:6]
    A = lambda s: tag + s
    B = lambda n: os.path.join(BIN, n)
    det = ["-H", "[100,125]", "-S", "[150,256]", "-V", "[100,256]", "-e", "3", "-d", "7"]
    procs = []
    reader = subprocess.Popen([B("oat-posi-cout"), A("pos")], stdout=subprocess.PIPE, text=True)
    if a.fused:
        procs.append(subprocess.Popen([B("oat-track-hip"), A("raw"), A("pos"), "-a", "0.01", "--area", "[20,100000]"] + det))
    else:
        procs.append(subprocess.Popen([B("oat-posidet-hip"), "hsv", A("hsv"), A("pos"), "-a", "[20,100000]"] + det))
        procs.append(subprocess.Popen([B("oat-framefilt-hip"), "col", A("filt"), A("hsv"), "-C", "HSV"]))
        procs.append(subprocess.Popen([B("oat-framefilt-hip"), "mog", A("raw"), A("filt"), "-a", "0.01"]))
    time.sleep(4.0)
    t0 = time.perf_counter()
    feeder = subprocess.Popen([B("oat-frameserve-raw"), A("raw"), "-f", raw, "--rows", str(a.rows), "--cols", str(a.cols),
                               "-n", str(a.frames)])
    out, _ = reader.communicate(timeout=600)
    el = time.perf_counter() - t0
    feeder.wait(timeout=60)
    for p in procs:
        p.wait(timeout=60)
    os.unlink(raw)
    subprocess.run([B("oat-clean-hip"), A("raw"), A("filt"), A("hsv"), A("pos")], capture_output=True)
    n = len([l for l in out.splitlines() if l.strip()])
    ok = sum('"pos_ok":true' in l for l in out.splitlines())
    print(f"{'fused oat-track-hip' if a.fused else '3-process chain'} {a.cols}x{a.rows}: {n} tokens in {el:.2f} s = "
          f"{n / el:.1f} fps ({ok} valid positions)")


if __name__ == "__main__":
    main()
