#!/usr/bin/env python3
"""Wall-clock throughput of the drop-in PROCESS pipeline over shared memory (the reference's own
perf methodology, test/perf/*.sh: `time` a free-running frame server with the consumer chain attached).

    python tools/pipeline_fps.py [--rows 480 --cols 640] [--frames 1000] [--fused] [--cameras N] [--ring D]

--cameras N (with --fused): N free-running frame servers -> ONE batched, pipelined oat-track-hip -> N readers;
the reported rate is the aggregate over all cameras.
"""
import argparse
import os
import re
import subprocess
import sys
import time
import uuid

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "build", "bin")


def node_pin(node):
    """preexec_fn: the child runs on the CPUs of NUMA node `node` (its first-touch allocations land there)."""
    if node == -2:
        from oat_amd import ffi
        node = ffi.load().oatgpu_device_numa_node(0)
    if node is None or node < 0:
        return None
    try:
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
    except OSError:
        return None
    return lambda: os.sched_setaffinity(0, cpus)


def batched(a):
    from oat_amd.synth import SyntheticStream
    tag = "oat_p_" + uuid.uuid4().hex[:6]
    B = lambda n: os.path.join(BIN, n)
    n = a.cameras
    srcs = [f"{tag}raw{s}" for s in range(n)]
    snks = [f"{tag}pos{s}" for s in range(n)]
    raws = []
    for s in range(n):
        st = SyntheticStream(a.rows, a.cols, s, n_discs=2)
        raw = f"/dev/shm/oat_pipe_{uuid.uuid4().hex[:8]}.raw"
        np.stack([st.frame(t, with_discs=t > 0) for t in range(8)]).tofile(raw)
        raws.append(raw)
    # readers write to FILES: n pipes drained one after the other would fill up and stall the whole pipeline
    outs = [open(f"/dev/shm/{x}.out", "w+") for x in snks]
    readers = [subprocess.Popen([B("oat-posi-cout"), x], stdout=o, text=True) for x, o in zip(snks, outs)]
    terr = open(f"/dev/shm/{tag}.trk.err", "w+") if a.timing else None
    tracker = subprocess.Popen((a.tracker_prefix.split() if a.tracker_prefix else []) +
                               [B("oat-track-hip"), ",".join(srcs), ",".join(snks), "-a", "0.01", "--area", "[20,100000]",
                                "-H", "[100,125]", "-S", "[150,256]", "-V", "[100,256]", "-e", "3", "-d", "7",
                                "--ring", str(a.ring)] + (["--stage-copy", a.stage_copy] if a.stage_copy else []) + (["--timing"] if a.timing else []) + a.tracker_extra.split(),
                               stderr=terr)
    time.sleep(12.0 if a.tracker_prefix else 4.0)
    t0 = time.perf_counter()
    feeders = [subprocess.Popen([B("oat-frameserve-raw"), srcs[s], "-f", raws[s], "--rows", str(a.rows), "--cols",
                                 str(a.cols), "-n", str(a.frames)], preexec_fn=node_pin(a.feeder_node)) for s in range(n)]
    tokens = ok = 0
    for r in readers:
        r.wait(timeout=300)
    el = time.perf_counter() - t0
    for o in outs:
        o.seek(0)
        out = o.read()
        o.close()
        os.unlink(o.name)
        tokens += len([l for l in out.splitlines() if l.strip()])
        ok += sum('"pos_ok":true' in l for l in out.splitlines())
    for f in feeders:
        f.wait(timeout=60)
    tracker.wait(timeout=60)
    for raw in raws:
        os.unlink(raw)
    subprocess.run([B("oat-clean-hip")] + srcs + snks, capture_output=True)
    steady = ""
    if terr:
        terr.seek(0)
        txt = terr.read()
        terr.close()
        os.unlink(terr.name)
        for l in txt.splitlines():
            if "per round" in l or "per step (ms)" in l:
                print(l)
                m = re.search(r"steady ([0-9.]+) fps", l)
                if m:
                    steady = f"; the tracker's own clock, rounds 17..: {m.group(1)} fps aggregate"
            elif l.strip() and "Exiting" not in l:
                print(l, file=sys.stderr)
    print(f"batched oat-track-hip, {n} cameras x {a.cols}x{a.rows}, ring {a.ring}: {tokens} tokens in {el:.2f} s = "
          f"{tokens / el:.1f} fps aggregate, {tokens / el / n:.1f} per camera ({ok} valid positions){steady}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--cols", type=int, default=640)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--fused", action="store_true")
    ap.add_argument("--cameras", type=int, default=1)
    ap.add_argument("--ring", type=int, default=2)
    ap.add_argument("--tracker-prefix", default="", help="command to run oat-track-hip under, e.g. 'rocprofv3 --memory-copy-trace -d /tmp/mc -o r --'")
    ap.add_argument("--tracker-extra", default="", help="more oat-track-hip options, e.g. '--gpu-index 0 --ingest-root 0' (the RCCL scatter form)")
    ap.add_argument("--timing", action="store_true", help="oat-track-hip --timing: where the tracker's loop spends its wall clock")
    ap.add_argument("--stage-copy", default="", choices=["", "dma", "kernel"], help="oat-track-hip --stage-copy (oatgpu_set_stage_copy)")
    ap.add_argument("--feeder-node", type=int, default=-1,
                    help="keep the frame servers (and so, by first touch, the shared-memory frames) on the CPUs of this NUMA "
                         "node; -2 = the node GPU 0 hangs off (oatgpu_device_numa_node)")
    a = ap.parse_args()
    if a.cameras > 1:
        return batched(a)
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
