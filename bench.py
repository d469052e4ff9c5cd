#!/usr/bin/env python3
"""bench.py -- throughput of the fused hot path (mog + hsv + erode/dilate + blob) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one frame for every camera stream a rank owns, through the whole
fused chain (MOG2 update, setTo, BGR2HSV, inRange, erode, dilate, labelling,
contour sums, selection, result hand-off).  Input frames are synthetic, uchar3,
and already resident in HBM when the timed region starts.  Streams are
independent, so N > 1 shards streams over ranks with no data-path collective
("weak" scaling: per-GPU work fixed); rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs):
    4k1      1 x 3840x2160 per GPU, erode 7 dilate 7 (configs[4]; the default: the config the
             north-star roofline target is stated on)
    1080p16  16 x 1920x1080 streams batched per GPU  (configs[2]; configs[3] at N=8 is 8/GPU)
    1080p8   8 x 1920x1080 per GPU                   (configs[3] shard)
    1080p1   1 x 1920x1080 stream per GPU            (configs[1])
    vga1     1 x 640x480                             (configs[0] shape)

Output.  Rank 0 prints ONE compact JSON line on stdout (<= 4 KB, numbers only: slim_line() below) -- the contract's
keys plus
    roofline      the dominant kernel k_mog_fused against the 8 TB/s HBM peak on the leg where its 205
                  algorithmic B/px really move (`--dense-model` input at 4K, run inside this process):
                  achieved / frac <= 1, frac_one_frame (one frame a launch: SURVEY 8d verbatim), traffic = PMC
                  bytes per launch of that same leg; frac_benched / waste_ratio for the benched (sparse) workload
    cpu_baseline  the oracle chain (a port of the reference's CPU path) on this host's cores
    extra_workloads  {name: fps} for 1080p16, 1080p8, 1080p1 and vga1 measured the same way (shorter runs)
    stage_ms, latency_us  device time of a frame through both halves; enqueue -> result-ready latency
    partition     which global streams every rank owned and its gate verdict (N > 1: both gates on EVERY rank)
    scatter_ingest  N > 1: the same hot path fed through the stream->rank scatter from rank 0 (RCCL send/recv)
-- and writes EVERYTHING it measured (notes, thread-split tables, audits, PMC detail, timing blocks) to
bench_detail.json beside this file and, as one line, to stderr.
`--gpus N` IS an N-rank run: without WORLD_SIZE in the environment the script re-executes itself through
`python -m torch.distributed.run --nproc-per-node N` (one rank per device; fewer devices than ranks is an error with the
"nccl" backend); under a launcher whose world differs from --gpus it refuses to run.
The PMC numbers come from rocprofv3 child processes of this very run (--no-pmc to skip them).
"""
import argparse
import json
import os
import re
import platform
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_PIXEL = 205           # 3 B BGR + 101 B model read + 101 B model write
FRAME_BYTES_PER_PIXEL = 3       # ... of which the frame
MODEL_BYTES_PER_PIXEL = 202     # ... and the model, read and written

WORKLOADS = {
    "1080p1": dict(rows=1080, cols=1920, streams=1, erode=3, dilate=7),
    "qhd1": dict(rows=1440, cols=2560, streams=1, erode=5, dilate=7),       # (lab sizes: not in extra_workloads)
    "1080p2": dict(rows=1080, cols=1920, streams=2, erode=3, dilate=7),
    "1080p3": dict(rows=1080, cols=1920, streams=3, erode=3, dilate=7),
    "1080p4": dict(rows=1080, cols=1920, streams=4, erode=3, dilate=7),
    "1080p8": dict(rows=1080, cols=1920, streams=8, erode=3, dilate=7),
    "1080p16": dict(rows=1080, cols=1920, streams=16, erode=3, dilate=7),
    "4k1": dict(rows=2160, cols=3840, streams=1, erode=7, dilate=7),
    "vga1": dict(rows=480, cols=640, streams=1, erode=3, dilate=7),
}
ALPHA = 0.01
RESTORE = 1                     # MOG2Invoker's `nmodes = nNewModes;` (oracle/mog2.c "Mode count"); 0 = the other reading
AREA = (20.0, 1e5)
RING = 8
AGE = 600                       # frames a model has seen before anything is warmed up or timed (see timed_run)
MIN_TIMED_MS = 50.0             # below this a timed region says little: flagged in the line


_T_PROCESS = time.perf_counter()


def log(*a):
    print(f"[+{time.perf_counter() - _T_PROCESS:6.1f}s]", *a, file=sys.stderr, flush=True)


DETAIL_PATH = os.path.join(ROOT, "bench_detail.json")
SLIM_LIMIT = 4096               # bytes of the ONE stdout line (a 21 KB line was not parsed by the driver)


def _sig(x, n=5):
    """Numbers of the stdout line carry n significant digits (the detail file has them in full)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        x = float(x)
    except (TypeError, ValueError):
        return None
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float(f"{x:.{n}g}")


def _dig(d, *path):
    for k in path:
        if not isinstance(d, dict) or d.get(k) is None:
            return None
        d = d[k]
    return d


def slim_line(full):
    """The ONE stdout line, built from everything the run measured (`full`, which goes to bench_detail.json and to
    stderr): the contract's keys plus compact numeric blocks, no prose.  Pure function of `full` (tests/test_bench_line.py
    builds it from canned dicts and holds it to SLIM_LIMIT bytes)."""
    r = full.get("roofline") or {}
    cfg = full.get("config") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                      "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    for k in ("value", "ms_per_step"):
        line[k] = _sig(line[k], 7)
    line["value_mean"] = _sig(full.get("value_mean"), 7)
    line["config"] = {"workload": cfg.get("name"), "streams_per_gpu": cfg.get("streams_per_gpu"), "rows": cfg.get("rows"),
                      "cols": cfg.get("cols"), "erode": cfg.get("erode"), "dilate": cfg.get("dilate"),
                      "learning_rate": cfg.get("learning_rate"), "mog_mixtures": 5,
                      "frames_per_launch": _sig(cfg.get("frames_per_launch"), 3),
                      "model_age_frames": cfg.get("model_age_frames"), "parallelism": cfg.get("parallelism")}
    line["roofline"] = {
        "kernel": r.get("kernel"), "bound": r.get("bound"), "peak": r.get("peak"), "unit": r.get("unit"),
        "achieved": _sig(r.get("achieved")), "frac": _sig(r.get("frac"), 4),
        "frac_one_frame": _sig(r.get("frac_one_frame"), 4), "frac_benched": _sig(r.get("frac_benched"), 4),
        "bytes_per_launch": _sig(r.get("bytes_per_launch"), 8), "avg_launch_ms": _sig(r.get("avg_launch_ms")),
        "avg_launch_ms_one_frame": _sig(_dig(r, "one_frame_a_launch", "avg_launch_ms")),
        "traffic": _sig(r.get("traffic"), 8), "waste_ratio": _sig(r.get("waste_ratio"), 4),
        "measured_stream_copy_GBps": _sig(r.get("measured_stream_copy_GBps")),
        "leg": "4k1 dense (5 live modes a pixel)" if r.get("frac") is not None else None,
        "benched_launch_ms": _sig(_dig(r, "benched_workload", "avg_launch_ms")),
        "benched_traffic": _sig(_dig(r, "benched_workload", "traffic"), 8),
        "frac_benched_source": r.get("frac_benched_source"),
        "k1_ms_ranks": ([_sig(_dig(r, "k_mog_fused_ms_ranks", "min"), 4), _sig(_dig(r, "k_mog_fused_ms_ranks", "max"), 4)]
                        if r.get("k_mog_fused_ms_ranks") else None)}
    cb = full.get("cpu_baseline")
    if cb:
        host = cb.get("host") or {}
        line["cpu_baseline"] = {"value": _sig(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"),
                                "kind": cb.get("kind"), "method": cb.get("method"),
                                "sample": cb.get("sample_short") or str(cb.get("sample", ""))[:100],
                                "value_1thread": _sig(cb.get("value_1thread")), "nproc": host.get("nproc"),
                                "physical_cores": host.get("physical_cores"), "cpu_model": host.get("cpu_model"),
                                "other_sizes": {k: _sig(v.get("value")) for k, v in (cb.get("other_sizes") or {}).items()}}
    else:
        line["cpu_baseline"] = None
    line["value_one_frame_a_launch"] = _sig(full.get("value_one_frame_a_launch"), 6)
    line["value_default_learning_rate_0"] = _sig(full.get("value_default_learning_rate_0"), 6)
    line["fps_per_gpu"] = _sig(full.get("fps_per_gpu"), 7)
    line["stage_ms"] = {k: _sig(v, 4) for k, v in (full.get("stage_ms") or {}).items()}
    line["latency_us"] = {k: _sig(v, 4) for k, v in (full.get("latency_us") or {}).items()} or None
    line["parity"] = full.get("parity")
    line["positions_found"] = full.get("positions_found")
    line["positions_expected"] = full.get("positions_expected")
    line["positions_with_target"] = full.get("positions_with_target")
    ew = full.get("extra_workloads")
    line["extra_workloads"] = ({k: _sig(v.get("value"), 6) for k, v in ew.items()} if ew else None)
    line["extra_parity"] = ({k: v.get("parity") for k, v in ew.items() if v.get("parity") != "ok"} or "ok") if ew else None
    pl = full.get("pipeline")
    if pl:
        line["pipeline"] = {"framefilt_mog_1MP_fps": _sig(_dig(pl, "framefilt_mog_1MP", "fps")),
                            "posidet_hsv_1MP_fps": _sig(_dig(pl, "posidet_hsv_1MP", "fps")),
                            "track_1080p_latency_us_p50": _sig(_dig(pl, "track_1080p_latency", "free_running", "latency_us", "p50")),
                            "track_1080p_latency_us_p99": _sig(_dig(pl, "track_1080p_latency", "free_running", "latency_us", "p99")),
                            "track_8x1080p_fps": _sig(_dig(pl, "track_8x1080p", "fps_aggregate"))}
    else:
        line["pipeline"] = None
    pt = full.get("partition") or {}
    line["partition"] = {"rule": "stream s -> rank s // ceil(S/N)", "streams_total": pt.get("streams_total"),
                         # [rank, device, first stream, one past the last, gate verdict, k_mog_fused ms]
                         "ranks": [[q.get("rank"), q.get("device"), (q.get("streams") or [None, None])[0],
                                    (q.get("streams") or [None, None])[1], q.get("parity"), _sig(q.get("k_mog_fused_ms"), 4)]
                                   for q in (pt.get("per_rank") or [])]}
    line["rccl"] = full.get("rccl")
    sc = full.get("scatter_ingest")
    keep = ("fps", "ms_per_step", "bytes_per_peer", "parity", "steps", "depth", "backend", "error", "workload")
    line["scatter_ingest"] = ({k: (_sig(v) if not isinstance(v, str) else v) for k, v in sc.items() if k in keep} if sc else None)
    if sc and sc.get("also"):                      # N > 1: the other workload's shard through the same scatter
        line["scatter_ingest"]["also"] = {k: (_sig(v) if not isinstance(v, str) else v) for k, v in sc["also"].items() if k in keep}
    line["device_open_retries"] = full.get("device_open_retries")
    line["timed_region_ms"] = _sig(full.get("timed_region_ms"))
    line["bench_wall_s"] = _sig(full.get("bench_wall_s"), 4)
    line["detail"] = "bench_detail.json"
    return line


def emit(full):
    """Rank 0: everything to bench_detail.json and stderr, the compact line -- and only it -- to stdout."""
    slim = slim_line(full)
    text = json.dumps(slim, separators=(",", ":"))
    if len(text) > SLIM_LIMIT:          # never hand the driver a line it cannot parse: drop the optional blocks, largest first
        for k in ("pipeline", "extra_workloads", "latency_us", "scatter_ingest", "partition"):
            slim[k] = None
            text = json.dumps(slim, separators=(",", ":"))
            if len(text) <= SLIM_LIMIT:
                break
    try:
        with open(DETAIL_PATH, "w") as f:
            json.dump(full, f, indent=1, default=str)
    except OSError as e:
        log("bench_detail.json not written:", e)
    log("bench detail: " + json.dumps(full, default=str))
    print(text, flush=True)
    return text


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def spawn_ranks_if_asked(args, argv):
    """`bench.py --gpus N` IS an N-rank run (VERDICT r04 missing-2).  Launched plainly (no WORLD_SIZE) with N > 1 the
    script replaces itself by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`,
    one rank per device; with the "nccl" (= RCCL) backend fewer visible devices than ranks is an error (rc 3), never a
    silent N = 1.  Under a launcher (WORLD_SIZE set) the world must equal --gpus.  Returns the world size."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        world = int(env_world)
        if args.gpus is not None and args.gpus != world:
            log(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): refusing to run a "
                f"mislabelled job")
            sys.exit(2)
        return world
    n = args.gpus or 1
    if n < 1:
        log("bench.py: --gpus must be >= 1")
        sys.exit(2)
    if n == 1:
        return 1
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < 1 or (args.backend == "nccl" and ndev < n):
        log(f"bench.py: --gpus {n} needs {n} visible device(s), found {ndev} (backend {args.backend}: one rank per device; "
            f"`--backend gloo` lets ranks share a device for control-flow smoke tests)")
        sys.exit(3)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    log("bench.py: starting", n, "ranks:", " ".join(cmd))
    sys.stdout.flush()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def open_device_with_retry(dev, rank, attempts=6):
    """First touch of the device by this rank.  The ranks of a job open a freshly booted device in the same instant, and the
    first runtime call of one of them can fail transiently (seen once in six fresh-box runs of the 8-rank test, r04): bounded
    retry with back-off, every failure printed with its text, the count carried into the line -- tolerated and VISIBLE, never
    hidden by re-running the job."""
    n = 0
    for a in range(attempts):
        try:
            torch.cuda.set_device(dev)
            torch.empty(8, device=dev)       # an allocation creates the context.  NOT a kernel plus a read-back: `.sum().item()`
            torch.cuda.synchronize(dev)      # here, before the process was pinned and the library had made its streams, cost the
            return n                         # host-bound workloads 20-40 % with identical kernel durations (vga1 78 k -> 48 k fps, one
                                             # 1080p stream 50 k -> 40 k: profiles/r07b_small_workloads_vs_r04.txt, r07d_...; mechanism
                                             # not established -- the library's streams do run side by side either way, r07v / r07w)
        except Exception as e:
            n += 1
            log(f"[rank {rank}] opening {dev} failed (attempt {a + 1}/{attempts}): {type(e).__name__}: {str(e)[-300:]}")
            if a + 1 == attempts:
                raise
            time.sleep(0.05 * (1 << a))
    return n


def pin_to_gpu_node(dev_index):
    """Keep this process on the CPUs of the NUMA node the GPU hangs off.  The small workloads are bound by the
    host's launch calls (a 1080p step is ~35 us of HIP API work), and after the many-threaded oracle gates the
    scheduler may leave the driving thread on the far socket: the same leg then measured 15.8 k instead of 27.9 k
    fps.  Best effort; returns the node or None."""
    try:
        import glob
        cards = sorted(glob.glob("/sys/class/drm/card*/device/numa_node"))
        amd = [c for c in cards if open(os.path.join(os.path.dirname(c), "vendor")).read().strip() == "0x1002"]
        node = int(open(amd[dev_index % len(amd)]).read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        os.sched_setaffinity(0, cpus & os.sched_getaffinity(0) or cpus)
        return node
    except Exception:
        return None


def make_hotpath(wl, device, ring_depth=RING, dense=False, n_streams=None):
    import oat_amd
    from oat_amd.synth import disc_hsv_window
    win = dict(h_thresh=(0, 256), s_thresh=(0, 256), v_thresh=(255, 256)) if dense else disc_hsv_window()
    return oat_amd.HotPath(wl["rows"], wl["cols"], n_streams=n_streams or wl["streams"], adaptation_coeff=ALPHA,
                           erode=wl["erode"], dilate=wl["dilate"], area=AREA, device=device,
                           ring_depth=ring_depth, mog_restore_nmodes=RESTORE, **win)


def oracle_params(wl):
    import oracle_lib as O
    return O.hsv_params(h_lo=100, h_hi=125, s_lo=150, s_hi=256, v_lo=100, v_hi=256, erode=wl["erode"],
                        dilate=wl["dilate"], min_area=AREA[0], max_area=AREA[1])


def oracle_mog(wl):
    import oracle_lib as O
    return O.Mog2(wl["rows"], wl["cols"], 3, params=dict(restore_nmodes=RESTORE))


def physical_cores():
    """Distinct (socket, core) pairs of this host (the hardware threads of one core share its ALUs and caches)."""
    try:
        seen, phys = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                seen.add((phys, line.split(":")[1].strip()))
        if seen:
            return len(seen)
    except OSError:
        pass
    return os.cpu_count() or 1


THREAD_SHARE = 1                # ranks of this job on this host (N > 1: every rank runs its own gates; they share the cores)


def host_threads():
    # the port's row workers are a persistent pool since r03 (oracle/pool.c): one per physical core
    return max(1, min(physical_cores() // max(THREAD_SHARE, 1), 256))


def parity_gate(wl, device, frames_seq):
    """masks pixel-exact and centroids identical vs the oracle, on a short fresh run
    (the oracle only CHECKS here; it is never the thing measured)."""
    import oracle_lib as O
    hp = make_hotpath(wl, device, ring_depth=2, n_streams=1)
    orc = oracle_mog(wl)
    p = oracle_params(wl)
    try:
        for t, f in enumerate(frames_seq):
            got = hp.track([f])[0]
            want, thr = O.chain_step(orc, f, ALPHA, p, nthreads=host_threads())
            if not (hp.read_mask(1) == thr).all():
                return f"mask mismatch at frame {t}"
            if got.position_valid != want["valid"]:
                return f"valid mismatch at frame {t}"
            if want["valid"] and (abs(got.x - want["x"]) > 1e-4 or abs(got.y - want["y"]) > 1e-4):
                return f"centroid mismatch at frame {t}"
    finally:
        hp.close()
    return "ok"


def measured_run_gate(wl, state, frames_of_step, got_positions, tag=""):
    """SURVEY.md 8d 'parity gates run with every benchmark': the positions the TIMED run itself produced for one
    camera stream against the oracle chain.  The model is AGED on the device first (hundreds to thousands of
    frames: a young model is a transient, not the workload), so the oracle does not replay the history: it takes
    over the device's model as exported right after the ageing (`state` = HotPath.mog_state(): counters, weights,
    variances, means, frame count) and runs the very frames that followed -- warm-up, then the timed steps.
    frames_of_step: host frames of that stream from the hand-over on; got_positions: (index into them, Position2D)."""
    import oracle_lib as O
    orc = oracle_mog(wl)
    nm, w, v, m, nframes = state
    orc.set_state(nm, w, v, m, nframes)
    p = oracle_params(wl)
    want = [O.chain_step(orc, f, ALPHA, p, nthreads=host_threads())[0] for f in frames_of_step]
    for t, g in got_positions:
        w_ = want[t]
        if g.position_valid != w_["valid"]:
            return f"{tag}valid mismatch at run frame {t}"
        if w_["valid"] and ((g.a00, g.a10, g.a01) != (w_["a00"], w_["a10"], w_["a01"]) or
                            abs(g.x - w_["x"]) > 1e-4 or abs(g.y - w_["y"]) > 1e-4):
            return f"{tag}centroid mismatch at run frame {t}"
    return "ok"


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_time_chain(wl, frames_seq, nthreads, budget_s, max_frames):
    """frames/s of the oracle chain (a port of the reference's CPU path) with `nthreads` row workers."""
    import oracle_lib as O
    orc = oracle_mog(wl)
    p = oracle_params(wl)
    O.chain_step(orc, frames_seq[0], ALPHA, p, nthreads=nthreads)      # frame 1 (model init), untimed
    O.chain_step(orc, frames_seq[1 % len(frames_seq)], ALPHA, p, nthreads=nthreads)
    n, t0 = 0, time.perf_counter()
    while True:
        O.chain_step(orc, frames_seq[(n + 2) % len(frames_seq)], ALPHA, p, nthreads=nthreads)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= max_frames:
            break
    return n / el, n, el


def cpu_time_pipeline(wl, frames_seq, t_front, t_mid, pipelined, budget_s, max_frames):
    """frames/s of the oracle chain through oracle/pipeline.c: pipelined = the reference's three concurrent stages
    (framefilt mog | framefilt col + posidet up to the morphology | findContours), else stage after stage; also the
    milliseconds per frame every stage was busy."""
    import oracle_lib as O
    orc = oracle_mog(wl)
    p = oracle_params(wl)
    L = len(frames_seq)
    O.pipeline_run(orc, frames_seq, 0, 2, ALPHA, p, t_front, t_mid, pipelined, keep=False)      # frame 1 (model init) + one more, untimed
    el, busy, _ = O.pipeline_run(orc, frames_seq, 2 % L, 4, ALPHA, p, t_front, t_mid, pipelined, keep=False)   # calibration
    n = int(max(4, min(max_frames, budget_s / max(el / 4, 1e-6))))
    el, busy, _ = O.pipeline_run(orc, frames_seq, 6 % L, n, ALPHA, p, t_front, t_mid, pipelined, keep=False)
    return n / el, n, el, dict(mog=busy[0] / n * 1e3, col_inrange_morph=busy[1] / n * 1e3, contours=busy[2] / n * 1e3)


def cpu_baseline_one(wl, frames_seq, budget_s):
    """The oracle timed on this host, bounded sample.  Stage after stage per frame (rounds 1-3's `value`): on 32 threads,
    on all physical cores, on every hardware thread, and on one thread.  And with the reference's STAGE PIPELINING
    (r04; FrameFilter.cpp:59-98, PositionDetector.cpp:58-99: three component processes work on consecutive frames at the
    same time, throughput 1 / max(stage) instead of 1 / sum(stage)): a few splits of the row workers between the MOG2
    stage and the colour / inRange / morphology stage, contour following on a thread of its own.  `value` is the best of
    everything -- the CPU's real best on this host."""
    phys, hw = physical_cores(), os.cpu_count() or 1
    legs = {}
    for nt in sorted({min(32, hw), phys, hw}):
        fps, n, el = cpu_time_chain(wl, frames_seq, nt, budget_s / 8, 2000)
        legs[nt] = dict(value=fps, frames=n, seconds=el)
    f1, n1, el1 = cpu_time_chain(wl, frames_seq, 1, budget_s / 8, 500)
    best = max(legs, key=lambda k: legs[k]["value"])
    # per-stage milliseconds of the best sequential leg, and the pipelined legs
    _, _, _, seq_stage = cpu_time_pipeline(wl, frames_seq, best, best, False, budget_s / 12, 300)
    pipe = {}
    half, quarter = max(1, best // 2), max(1, best // 4)
    for tf, tm in sorted({(quarter, half + quarter), (half, half), (half, best), (half, best + half), (best, half), (best, best),
                          (best + half, half), (min(2 * best, max(hw // 2, 1)), best)}):
        fps, n, el, st = cpu_time_pipeline(wl, frames_seq, tf, tm, True, budget_s / 12, 2000)
        pipe[f"{tf}+{tm}+1"] = dict(value=fps, frames=n, seconds=el, stage_ms=st)
    # the two best splits once more (run-to-run spread of one split on one box: 78-112 fps at 4K, r05q): `value` is the best seen
    for k in sorted(pipe, key=lambda k: -pipe[k]["value"])[:2]:
        tf, tm = (int(x) for x in k.split("+")[:2])
        fps, n, el, st = cpu_time_pipeline(wl, frames_seq, tf, tm, True, budget_s / 12, 2000)
        pipe[k]["repeat"] = fps
        if fps > pipe[k]["value"]:
            pipe[k].update(value=fps, frames=n, seconds=el, stage_ms=st)
    pbest = max(pipe, key=lambda k: pipe[k]["value"])
    v_seq, v_pipe = legs[best]["value"], pipe[pbest]["value"]
    return dict(value=max(v_seq, v_pipe), unit="frames/s", cores=(best if v_seq >= v_pipe else sum(int(x) for x in pbest.split("+"))),
                kind="port", method="stage-pipelined" if v_pipe > v_seq else "stage after stage",
                value_sequential=v_seq, value_pipelined=v_pipe,
                sample=f"{legs[best]['frames']} frames of one {wl['cols']}x{wl['rows']} stream, {legs[best]['seconds']:.1f} s, oracle chain stage "
                       f"after stage (MOG2, HSV, inRange, morphology: rows over a persistent pool of {best} workers; contour following 1 "
                       f"thread, as OpenCV's findContours); and {pipe[pbest]['frames']} frames, {pipe[pbest]['seconds']:.1f} s, with the "
                       f"reference's three stages concurrent on consecutive frames (threads mog+col/morph+contours = {pbest})",
                sample_short=f"{max(legs[best]['frames'], pipe[pbest]['frames'])} frames of one {wl['cols']}x{wl['rows']} stream "
                             f"per leg, oracle chain, best of {len(legs)} sequential + {len(pipe)} stage-pipelined thread splits",
                by_threads={str(k): v["value"] for k, v in legs.items()},
                stage_ms_sequential=seq_stage,
                pipelined={k: dict(value=v["value"], stage_ms=v["stage_ms"]) for k, v in pipe.items()},
                pipelined_best=pbest,
                value_1thread=f1, sample_1thread=f"{n1} frames, {el1:.1f} s, 1 thread")


def cpu_time_streams(wl, n_streams, frames_seq, t_front, t_mid, budget_s):
    """Aggregate frames/s of n_streams INDEPENDENT cameras on this host at once -- the reference's N-camera shape, one
    component chain per camera (examples/two-gige/two-gige.sh:7-8): every stream its own model and its own three-stage
    pipeline (oracle/pipeline.c) with t_front + t_mid + 1 threads, all streams concurrently -- so the contour followers
    of different streams run in parallel (`cpu_all`)."""
    import threading
    import oracle_lib as O
    p = oracle_params(wl)
    L = len(frames_seq)
    mogs = [oracle_mog(wl) for _ in range(n_streams)]
    for m in mogs:
        O.pipeline_run(m, frames_seq, 0, 2, ALPHA, p, t_front, t_mid, True, keep=False)
    el, _, _ = O.pipeline_run(mogs[0], frames_seq, 2 % L, 3, ALPHA, p, t_front, t_mid, True, keep=False)   # one stream alone: calibration
    n = int(max(3, min(200, budget_s / max(el / 3 * n_streams / 1.5, 1e-6))))     # (the streams slow each other down)
    times = [0.0] * n_streams

    def one(i):
        times[i] = O.pipeline_run(mogs[i], frames_seq, 5 % L, n, ALPHA, p, t_front, t_mid, True, keep=False)[0]
    th = [threading.Thread(target=one, args=(i,)) for i in range(n_streams)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    return dict(value=n_streams * n / wall, unit="frames/s aggregate", streams=n_streams, frames_per_stream=n, seconds=wall,
                threads_per_stream=f"{t_front}+{t_mid}+1", fps_per_stream=n / max(times))


def cpu_baseline(name, frames_seq):
    """Benched workload (about 14 s) plus short samples of the other two sizes BASELINE.md section 2 asks for."""
    out = cpu_baseline_one(WORKLOADS[name], frames_seq, 18.0)
    out["host"] = dict(nproc=os.cpu_count(), physical_cores=physical_cores(), cpu_model=cpu_model_string(),
                       note="`cores` = row workers of the best leg; by_threads lists every leg (32 = round 2's cap, "
                            "physical cores, all hardware threads)")
    from oat_amd.synth import SyntheticStream
    others = {}
    for other in ("vga1", "1080p1", "4k1"):
        w = WORKLOADS[other]
        if (w["rows"], w["cols"]) == (WORKLOADS[name]["rows"], WORKLOADS[name]["cols"]):
            continue
        st = SyntheticStream(w["rows"], w["cols"], 0, n_discs=2)
        fr = [st.frame(t, with_discs=t > 0) for t in range(4)]
        r = cpu_baseline_one(w, fr, 6.0)
        others[other] = dict(value=r["value"], value_sequential=r["value_sequential"], value_pipelined=r["value_pipelined"],
                             value_1thread=r["value_1thread"], unit="frames/s", cores=r["cores"], method=r["method"],
                             by_threads=r["by_threads"], pipelined=r["pipelined"], stage_ms_sequential=r["stage_ms_sequential"],
                             sample=r["sample"])
    out["other_sizes"] = others
    # configs[3]'s per-GPU shard on the CPU: 8 x 1080p cameras at once (every stream its own pipeline, contours in parallel)
    try:
        w = WORKLOADS["1080p8"]
        st = SyntheticStream(w["rows"], w["cols"], 0, n_discs=2)
        fr = [st.frame(t, with_discs=t > 0) for t in range(4)]
        hw = os.cpu_count() or 1
        per = max(1, hw // (2 * w["streams"]) // 2)          # row workers per stage and stream: half the hardware threads in all
        out["multi_stream_1080p8"] = cpu_time_streams(w, w["streams"], fr, per, per, 5.0)
    except Exception as e:
        out["multi_stream_1080p8"] = dict(error=str(e)[-200:])
    return out


LAB_CALM = False
DENSE_NOISE = 5       # amplitude of the dense leg's per-frame noise (--dense-noise)
EARLY_BLOB = None     # None: the library's default (by shape: at most three streams, 4 MP a step and more); False / True: oatgpu_set_early_blob (--early-blob)
FUSION = 2            # frames per launch of the fused per-pixel kernel on the pipelined path (--fusion; oatgpu_set_fusion)


def make_pool_dense(rows, cols, ns, nframes, rank, dev):
    """The case where all 205 algorithmic B/px really move: every pixel cycles through FIVE well separated
    colours in a fixed order (own phase per pixel).  All five mixture modes stay live with near-equal weights, the
    mode that matches is the one matched longest ago -- the LAST slot -- so no pixel ever matches mode 0 (every
    lane loads all 25 planes), and bubbling it to the front rewrites every plane.  nframes must be a multiple of 5
    (the pool wraps).  The kernel's traffic audit confirms 104 B read + 101 B written per pixel."""
    assert nframes % 5 == 0
    g = torch.Generator(device=dev)
    g.manual_seed(0xD0 + rank)
    table = torch.tensor([[20, 30, 40], [90, 200, 60], [200, 60, 120], [240, 240, 230], [40, 130, 220]],
                         device=dev, dtype=torch.int16)
    phase = torch.randint(0, 5, (ns, rows, cols), device=dev, generator=g)
    pool = []
    for t in range(nframes):
        f = table[(phase + t) % 5] + torch.randint(-DENSE_NOISE, DENSE_NOISE + 1, (ns, rows, cols, 3), device=dev, dtype=torch.int16, generator=g)
        pool.append(f.clamp_(0, 255).to(torch.uint8).contiguous())
    return pool


def make_pool_device(rows, cols, ns, nframes, rank, dev):
    """[nframes] tensors of shape (ns, rows, cols, 3) uint8 on `dev`: gradient + fresh +-6 noise per frame,
    every 64th pixel flickering, two saturated discs per stream on Lissajous paths."""
    from oat_amd.synth import DISC_BGR
    g = torch.Generator(device=dev)
    g.manual_seed(0x0A7 + rank)
    yy = torch.arange(rows, device=dev, dtype=torch.float32).view(rows, 1)
    xx = torch.arange(cols, device=dev, dtype=torch.float32).view(1, cols)
    base = 110 + 30 * (xx / max(cols - 1, 1)) + 20 * (yy / max(rows - 1, 1))
    base = torch.stack([base, base + 6, base - 5], -1).to(torch.int16)            # (rows, cols, 3)
    flick = ((torch.arange(rows * cols, device=dev) % 64) == 17).view(rows, cols)
    if LAB_CALM:                                    # lab only (--lab-calm): no flickering pixel, the path every lane shares
        flick = torch.zeros_like(flick)
    rmin = max(4, min(rows, cols) // 40)
    cpu_rng = np.random.default_rng(0x0A7 + rank)
    discs = []
    for s in range(ns):
        ds = []
        for d in range(2):
            r = int(cpu_rng.integers(rmin, 2 * rmin + 1))
            ds.append(dict(r=r, col=DISC_BGR[d], ax=(cols / 2 - r - 24) * (0.55 + 0.15 * d),
                           ay=(rows / 2 - r - 24) * (0.5 + 0.2 * d),
                           fx=0.013 * (1 + d) + 0.002 * (s % 7), fy=0.017 * (1 + 0.5 * d) + 0.001 * (s % 5),
                           px=cpu_rng.uniform(0, 6.28), py=cpu_rng.uniform(0, 6.28)))
        discs.append(ds)
    # ONE allocation for the whole pool and no boolean-mask indexing (a `nonzero` = a device synchronisation each): eight ranks
    # sharing one device (the `--backend gloo` smoke form of an N-GPU run) spent TEN MINUTES here with 48 frames a rank -- 0.3 to
    # 18 s as plain processes, 1 000 s under the launcher, 1 s with 24 frames (profiles/r08_gpus8_startup.txt); the bytes are
    # the same as before.
    flick_add = [flick.unsqueeze(-1).to(torch.int16) * v for v in (-50, 60)]                   # (rows, cols, 1), by frame parity
    cols_dev = {tuple(c): torch.tensor(c, device=dev, dtype=torch.uint8) for c in DISC_BGR}
    store = torch.empty((nframes, ns, rows, cols, 3), device=dev, dtype=torch.uint8)
    for t in range(nframes):
        f = base.unsqueeze(0) + torch.randint(-6, 7, (ns, rows, cols, 3), device=dev, dtype=torch.int16, generator=g)
        f += flick_add[t & 1]
        torch.clamp(f, 0, 255, out=f)
        store[t].copy_(f)
        if t > 0:                                   # frame 0 (model initialisation) has no discs
            for s in range(ns):
                for d in discs[s]:
                    cx = int(round(cols / 2 + d["ax"] * np.sin(d["fx"] * 9 * t + d["px"])))
                    cy = int(round(rows / 2 + d["ay"] * np.sin(d["fy"] * 9 * t + d["py"])))
                    r = d["r"]
                    y0, y1, x0, x1 = max(cy - r, 0), min(cy + r + 1, rows), max(cx - r, 0), min(cx + r + 1, cols)
                    m = ((xx[:, x0:x1] - cx) ** 2 + (yy[y0:y1] - cy) ** 2) <= r * r
                    sub = store[t, s, y0:y1, x0:x1]
                    sub.copy_(torch.where(m.unsqueeze(-1), cols_dev[tuple(d["col"])], sub))
    return [store[t] for t in range(nframes)]


class Leg:
    """One workload on one device: synthetic pool resident in HBM + a HotPath context; frames are consumed
    in pool order (frame 0 initialises the models), so the frame of every step is known afterwards."""

    def __init__(self, name, dev_index, rank, dense=False, pool=48, input_mode="device"):
        self.name, self.wl, self.dense = name, WORKLOADS[name], dense
        self.dev = torch.device("cuda", dev_index)
        wl = self.wl
        self.ns = wl["streams"]
        gen = make_pool_dense if dense else make_pool_device
        self.pool = gen(wl["rows"], wl["cols"], self.ns, pool, rank, self.dev)
        torch.cuda.synchronize()
        self.hp = make_hotpath(wl, dev_index, dense=dense)
        self.hp.set_fusion(FUSION)
        if EARLY_BLOB is not None:
            self.hp.set_early_blob(EARLY_BLOB)
        self.step = 0                    # frames consumed so far
        self.input_mode = input_mode
        self.host_pool = None
        if input_mode == "host":         # page-locked frames through the pipelined oatgpu_track_enqueue
            self.host_pool = [[f.numpy() for f in p.cpu().pin_memory()] for p in self.pool]
        elif input_mode == "host-sync":  # pageable frames through the synchronous oatgpu_track_batch
            self.host_pool = [[np.ascontiguousarray(f) for f in p.cpu().numpy()] for p in self.pool]

    def pool_index(self, step):          # step 0 = the initialisation frame
        return 0 if step == 0 else (step % len(self.pool))

    def init(self):
        self.hp.track_dev(self.pool[0].data_ptr())
        self.step = 1

    def age(self, min_frames, min_seconds=0.0):
        """Run the model to working age (untimed): at least min_frames frames and min_seconds of device time."""
        t0 = time.perf_counter()
        n = 0
        while n < min_frames or time.perf_counter() - t0 < min_seconds:
            k = max(1, min(200, min_frames - n)) if n < min_frames else 100
            self.run(k)
            n += k
        self.hp.synchronize()
        return n

    def export_models(self):
        """Every stream's model as the oracle takes it over (HotPath.mog_state())."""
        return [self.hp.mog_state(s) for s in range(self.ns)]

    def prepare(self, n):
        """Argument arrays of an n-step run, built outside the timed region."""
        import ctypes as C
        from oat_amd import ffi
        if self.host_pool is not None:
            return None
        idx = [self.pool_index(self.step + i) for i in range(n)]
        return ((C.c_void_p * n)(*[self.pool[i].data_ptr() for i in idx]), (ffi.Position * (n * self.ns))())

    def run(self, n, prepared=None, keep=False, done_s=None, enq_s=None):
        """n steps through the pipelined path; returns [n][ns] raw ffi.Position when keep.  done_s / enq_s: ctypes double
        arrays of n entries that receive the time each step's result was collected / each step was handed over
        (device-frame input only)."""
        import ctypes as C
        from oat_amd import ffi
        hp = self.hp
        lib, ctx, lr = hp.lib, hp.ctx, hp.learning_coeff_
        if self.host_pool is None:
            seq, out = prepared if prepared is not None else self.prepare(n)
            if done_s is not None:
                ffi.check(lib, ctx, lib.oatgpu_track_sequence_dev_latency(ctx, seq, n, lr, out, done_s, enq_s))
            else:
                ffi.check(lib, ctx, lib.oatgpu_track_sequence_dev(ctx, seq, n, lr, out))
            self.step += n
            return out if keep else None
        out = (ffi.Position * (n * self.ns))()
        psz = C.sizeof(ffi.Position)
        if self.input_mode == "host-sync":
            for i in range(n):
                fs = self.host_pool[self.pool_index(self.step + i)]
                ptrs = (ffi._u8p * self.ns)(*[ffi.u8(f) for f in fs])
                buf = (ffi.Position * self.ns).from_buffer(out, i * self.ns * psz)
                ffi.check(lib, ctx, lib.oatgpu_track_batch(ctx, ptrs, self.ns, lr, buf))
            self.step += n
            return out if keep else None
        ptrs = [(ffi._u8p * self.ns)(*[ffi.u8(f) for f in fs]) for fs in self.host_pool]
        outstanding = got = 0
        for i in range(n):
            if outstanding == RING:
                buf = (ffi.Position * self.ns).from_buffer(out, got * self.ns * psz)
                ffi.check(lib, ctx, lib.oatgpu_track_collect(ctx, buf))
                got += 1
                outstanding -= 1
            ffi.check(lib, ctx, lib.oatgpu_track_enqueue(ctx, ptrs[self.pool_index(self.step + i)], self.ns, lr))
            outstanding += 1
        while outstanding:
            buf = (ffi.Position * self.ns).from_buffer(out, got * self.ns * psz)
            ffi.check(lib, ctx, lib.oatgpu_track_collect(ctx, buf))
            got += 1
            outstanding -= 1
        self.step += n
        return out if keep else None

    def close(self):
        self.hp.close()
        self.pool = None


def spin_up(name, dev_index, rank, seconds):
    """Warm the DEVICE (clocks, allocator, first-launch costs) with a scratch context of the same workload right
    before the real one runs its W warm-up steps (the model export for the parity gate leaves the device idle
    for a moment, and the driver's default run times only a few milliseconds: on a cold device round 1 measured
    19 k instead of 28 k fps).  Not part of W or K; never touches the measured models."""
    leg = Leg(name, dev_index, rank + 1000, pool=4)      # (never the dense generator: any warm device will do)
    leg.init()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        leg.run(40)
        n += 40
    leg.hp.synchronize()
    leg.close()
    return n


CAL_STEPS = 50                  # steps of the (untimed) calibration run that sizes the timed region


def timed_run(leg, K, W, barrier, prof_every, age_frames, export=True, spin=0.0, spin_args=None, reduce_max=None,
              min_ms=None, isolated=False):
    """Model initialisation, ageing (untimed), [export of the aged models for the parity gate, device spin-up on a
    scratch context while the real one rests], W warm-up steps, then the timed region between barriers:
    R back-to-back BLOCKS of exactly K steps in one pipelined run, R chosen so that the region lasts >= min_ms
    (MIN_TIMED_MS) whatever --steps is.  Every block's time is taken from result to result (the moment step
    bK+K-1's result was collected minus the moment step bK-1's was: oatgpu_track_sequence_dev_timed), so blocks
    1..R-1 are K steps of the pipeline in steady state; block 0 also carries the pipeline's fill.  The reported
    step time is the MEDIAN block / K.  reduce_max: all-reduce(MAX) of a list of floats over the ranks (N > 1).
    Returns a dict."""
    import ctypes as C
    from oat_amd.components import Position2D
    hp = leg.hp
    min_ms = MIN_TIMED_MS if min_ms is None else min_ms
    tag = f"[{leg.name} r{os.environ.get('RANK', '0')}]"
    leg.init()
    aged = leg.age(age_frames) if age_frames > 0 else 0
    log(tag, f"aged {aged} frames")
    models = leg.export_models() if export else None
    handover = leg.step                 # first frame the oracle will see after taking the models over
    log(tag, "models exported" if export else "no export")
    if spin > 0 and spin_args:
        spin_up(*spin_args, spin)       # the export left the device idle: warm it again, on a scratch context
    if W:
        leg.run(W)
    # calibration (untimed, part of the warm-up as far as the model is concerned): how long does a step take?
    cal = CAL_STEPS
    timed_dev = leg.host_pool is None
    hp.synchronize()
    t0 = time.perf_counter()
    cal_done = (C.c_double * cal)() if timed_dev else None
    leg.run(cal, done_s=cal_done)
    hp.synchronize()
    t_cal = (time.perf_counter() - t0) / cal
    if timed_dev:                # the steady half of it: result to result, free of the pipeline's fill and drain
        t_cal = (cal_done[cal - 1] - cal_done[cal // 2 - 1]) / (cal - cal // 2)
    if reduce_max:
        t_cal = reduce_max([t_cal])[0]
    R = 1
    R_max = max(2, 200000 // max(K, 1))
    if timed_dev:
        # (sized by the calibration run's STEADY step time with a margin, and run again with more blocks should the region
        # still end below min_ms: with --steps 20 a region sized by the whole calibration run, fill and drain included,
        # came out at 43 ms -- VERDICT r04 weak-10)
        R = int(min(max(2, -(-1.15 * min_ms * 1e-3 // (t_cal * K)) + 1), R_max))
    attempts = skipped = 0
    while True:
        attempts += 1
        if attempts > 1:
            skipped += n                 # steps of a region that came out short: warm-up as far as the model and the gate go
        n = R * K
        hp.profile(prof_every if n < 64 else max(prof_every, 8))   # HIP events around K1 on every Nth step of the timed region
        hp.profile_reset()
        prepared = leg.prepare(n)
        done = (C.c_double * n)() if timed_dev else None
        enq = (C.c_double * n)() if timed_dev else None
        barrier()
        t0 = time.perf_counter()
        out = leg.run(n, prepared, keep=True, done_s=done, enq_s=enq)
        barrier()
        elapsed = time.perf_counter() - t0
        log(tag, f"timed region: {R} blocks x {K} steps in {elapsed * 1e3:.1f} ms (attempt {attempts})")
        prof = hp.profile_read()
        hp.profile(0)
        short = [elapsed * 1e3 < min_ms]
        if reduce_max:                   # every rank takes the same decision
            short = [reduce_max([1.0 if short[0] else 0.0])[0] > 0.0]
        if not timed_dev or not short[0] or attempts >= 3 or R >= R_max:
            break
        R = int(min(max(R + 1, R * 1.25 * min_ms / max(elapsed * 1e3, 1e-3) + 1), R_max))
    first_step = leg.step - n            # the timed region's first step; pool frame 0 (it initialises the models) carries no disc
    with_target = sum(1 for t in range(n) if leg.pool_index(first_step + t) != 0) * leg.ns
    if timed_dev:
        ends = [done[(b + 1) * K - 1] for b in range(R)]
        blocks = [ends[0]] + [ends[b] - ends[b - 1] for b in range(1, R)]
    else:
        blocks = [elapsed]
    region = [elapsed]
    local_steady = sorted(blocks[1:]) if len(blocks) > 1 else list(blocks)
    local_block = local_steady[len(local_steady) // 2]
    if reduce_max:
        blocks = reduce_max(blocks)
        region = reduce_max(region)
    steady = sorted(blocks[1:]) if len(blocks) > 1 else list(blocks)
    median = steady[len(steady) // 2] if len(steady) % 2 else 0.5 * (steady[len(steady) // 2 - 1] + steady[len(steady) // 2])
    iso = None
    if isolated:                 # K steps alone between two barriers: fill AND drain inside (what rounds 1-2 reported)
        prepared = leg.prepare(K)
        barrier()
        t0 = time.perf_counter()
        leg.run(K, prepared)
        barrier()
        iso = time.perf_counter() - t0
        if reduce_max:
            iso = reduce_max([iso])[0]
    ns = leg.ns
    G = min(K, 256)              # the gate replays at most this many timed steps
    positions = [[Position2D.from_c(out[t * ns + s]) for s in range(ns)] for t in range(G)]
    found = sum(1 for t in range(n) for s_ in range(ns) if out[t * ns + s_].valid == 1)
    lat = None
    if timed_dev and n >= 8:     # enqueue -> result collected, per frame, steady part of the region (the ring is kept full)
        l_ = sorted((done[i] - enq[i]) * 1e6 for i in range(min(n // 4, K), n))
        lat = dict(p50=l_[len(l_) // 2], p99=l_[min(len(l_) - 1, int(len(l_) * 0.99))], ring_depth=RING)
    return dict(block_s=median, local_block_s=local_block, blocks=blocks, n_blocks=R, region_s=region[0], steps_timed=n, isolated_block_s=iso,
                positions=positions, found=found, prof=prof, models=models, handover=handover, aged=aged,
                gate_offset=W + cal + skipped, step_s_calibration=t_cal, region_attempts=attempts,
                saturated_latency_us=lat, with_target=with_target)


def single_frame_latency(leg, n):
    """One frame at a time through oatgpu_track_batch_dev (hand-over -> result, nothing else in flight): what a camera-paced
    tracker waits for its position.  Continues the leg's model (untimed as far as `value` goes).  Microseconds."""
    import ctypes as C
    from oat_amd import ffi
    hp = leg.hp
    out = (ffi.Position * leg.ns)()
    ts = []
    for i in range(n):
        ptr = C.c_void_p(leg.pool[leg.pool_index(leg.step)].data_ptr())
        t0 = time.perf_counter()
        ffi.check(hp.lib, hp.ctx, hp.lib.oatgpu_track_batch_dev(hp.ctx, ptr, hp.learning_coeff_, out))
        ts.append((time.perf_counter() - t0) * 1e6)
        leg.step += 1
    ts = sorted(ts[n // 10:])
    return dict(p50=ts[len(ts) // 2], p99=ts[min(len(ts) - 1, int(len(ts) * 0.99))], frames=len(ts))


def k1_ms(prof):
    """HIP-event time of the K1 launch on its own stream, minus what an EMPTY event pair measures there
    (calibrated by the library): the kernel's duration as rocprofv3 reports it."""
    raw = prof["mog_ms"] / max(prof["steps"], 1)
    return max(raw - prof["event_pair_ms"], 1e-6), raw


def gates(leg, offset, K, positions, check_steps, models, handover):
    """Both parity gates for this leg's run, every stream of the rank.  offset = steps between the hand-over of the
    models and the first timed step (warm-up + calibration)."""
    wl = leg.wl
    first = [leg.pool[leg.pool_index(i)][0].cpu().numpy() for i in range(4)]
    res = parity_gate(wl, leg.dev.index, first)
    log(f"[{leg.name}] parity gate (fresh context, 4 frames, masks + centroids):", res)
    if res != "ok" or check_steps <= 0 or models is None:
        return res, None
    G = min(check_steps, K, len(positions))
    order = [leg.pool_index(handover + i) for i in range(offset + G)]
    for s in range(leg.ns):
        host = {i: leg.pool[i][s].cpu().numpy() for i in set(order)}
        res = measured_run_gate(wl, models[s], [host[i] for i in order], [(offset + i, positions[i][s]) for i in range(G)],
                                tag=f"stream {s}: ")
        if res != "ok":
            break
    log(f"[{leg.name}] measured-run gate ({G} timed steps x {leg.ns} stream(s) vs oracle from the aged model):", res)
    detail = (f"fresh-context masks+centroids on 4 frames; first {G} timed steps of every one of the {leg.ns} "
              f"stream(s) vs the oracle continuing from the device's exported model (age {handover} frames) over the "
              f"run's own frames: {res}")
    return res, detail


def audit(leg, steps=6):
    """The kernel's own traffic count over `steps` further (untimed) steps of this leg's model."""
    leg.hp.traffic_audit(True)
    leg.run(steps)
    t = leg.hp.traffic_read()
    leg.hp.traffic_audit(False)
    px = max(t["pixels"], 1)
    # `pixels` counts the pixels of every audited LAUNCH; a launch covers frames_per_launch frames (oatgpu_set_fusion),
    # so the *_B_per_px figures below are per pixel and launch -- per pixel and FRAME they are that / frames_per_launch
    return dict(steps=steps, pixels=t["pixels"], launches=t["launches"],
                frames_per_launch=steps * leg.ns * leg.wl["rows"] * leg.wl["cols"] / px,
                useful_read_B_per_px=t["lane_bytes_read"] / px, useful_write_B_per_px=t["lane_bytes_written"] / px,
                sector32_read_B_per_px=t["sector32_bytes_read"] / px, sector32_write_B_per_px=t["sector32_bytes_written"] / px,
                sector64_read_B_per_px=t["sector64_bytes_read"] / px, sector64_write_B_per_px=t["sector64_bytes_written"] / px)


def mode_histogram(leg):
    nm, w, _, _, _ = leg.hp.mog_state(0)
    k = w.shape[1]
    live = ((w != 0) & (np.arange(k)[None, :] < nm[:, None])).sum(1)
    return dict(stream=0, modes_used=np.bincount(nm, minlength=6)[:6].tolist(),
                live_modes=np.bincount(live, minlength=6)[:6].tolist(),
                mean_modes_used=float(nm.mean()), mean_live_modes=float(live.mean()))


# ------------------------------------------- N > 1: the other workload, every rank --

def multi_rank_extra(name, K, W, local_rank, rank, world, reduce_max, args):
    """N > 1: a SECOND gated leg on every rank -- BASELINE configs[3]'s shard (8 x 1080p a rank) beside configs[4]'s (one 4K
    stream a rank), or the other way round -- so that ONE `bench.py --gpus N` run of the driver yields both of north_star's
    sizes at N GPUs (N cameras with a chain each: examples/two-gige/two-gige.sh:7-8).  Same contract
    as the main leg: barrier + device synchronisation on both sides, blocks of exactly K steps, max over ranks, both parity
    gates on every rank for every stream of its shard.  Returns a dict on every rank; rank 0's holds the gathered ranks."""
    import torch.distributed as dist
    el = Leg(name, local_rank, rank, pool=24 if WORKLOADS[name]["streams"] >= 8 else args.pool)

    def barrier():
        el.hp.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
    try:
        er = timed_run(el, K, max(W, 5), barrier, prof_every=8 if K >= 64 else 1, age_frames=AGE, export=not args.no_parity,
                       spin=0.0, reduce_max=reduce_max)
        par = "skipped"
        if not args.no_parity:
            par, _ = gates(el, er["gate_offset"], K, er["positions"], min(args.check_steps, 16), er["models"], er["handover"])
        rec = dict(rank=rank, parity=par, found=er["found"], k_mog_fused_ms=k1_ms(er["prof"])[0],
                   frames_per_launch=er["prof"]["mog_frames"] / max(er["prof"]["steps"], 1),
                   ms_per_step_local=er["local_block_s"] / K * 1e3)
    except Exception as e:               # never let the second leg take the first one's line down
        log(f"[rank {rank}] extra workload {name} failed:", e)
        er, rec = None, dict(rank=rank, parity=f"error: {str(e)[-160:]}", found=0, k_mog_fused_ms=None)
    finally:
        el.close()
        torch.cuda.empty_cache()
    allrec = [None] * world
    dist.all_gather_object(allrec, rec)
    if er is None or any(q.get("k_mog_fused_ms") is None for q in allrec):
        return dict(name=name, value=None, parity=next(q["parity"] for q in allrec if q["parity"] != "ok"), per_rank=allrec)
    w_ = WORKLOADS[name]
    total = w_["streams"] * world
    bad = [q for q in allrec if q["parity"] != "ok"]
    k1s = [q["k_mog_fused_ms"] for q in allrec]
    return dict(name=name, value=total * K / er["block_s"], unit="frames/s", steps=K, blocks=er["n_blocks"], warmup=max(W, 5),
                streams_total=total, streams_per_gpu=w_["streams"], ms_per_step=er["block_s"] / K * 1e3,
                timed_region_ms=er["region_s"] * 1e3, model_age_frames=er["handover"],
                k_mog_fused_ms=max(k1s), k_mog_fused_ms_min=min(k1s), frames_per_launch=allrec[0]["frames_per_launch"],
                px_per_launch=w_["rows"] * w_["cols"] * w_["streams"],
                positions_found=sum(q["found"] for q in allrec), positions_expected=total * er["steps_timed"],
                parity="ok" if not bad else f"rank {bad[0]['rank']}: {bad[0]['parity']}", per_rank=allrec)


# -------------------------------------------------------- N > 1: scatter ingest --

def scatter_leg(name, world, rank, dev, backend, steps, reduce_max, depth=2, gate_steps=3):
    """N > 1, every rank: the hot path fed through the stream->rank scatter (north_star: "RCCL over xGMI only for the trivial
    stream-to-rank scatter").  ALL frames originate on rank 0 -- a camera host -- and travel through
    oat_amd.dist.FrameScatterPipe (send/recv per peer, `depth` slots, the hot path as consumer: the frames of step t+1 move
    while step t computes); each rank runs its own shard's chain on what arrived.  Fresh models; the first gate_steps + 1
    frames every rank RECEIVED are replayed through the oracle from scratch (masks' consequences: validity, exact contour
    sums, centroids) -- the gate checks the transport and the chain together; the steps behind them are timed between two
    barriers, max over ranks.  Returns a dict on every rank (rank 0's carries the gathered verdicts)."""
    import torch.distributed as dist
    import oracle_lib as O
    from oat_amd.dist import FrameScatterPipe
    from oat_amd.components import Position2D  # noqa: F401
    wl = WORKLOADS[name]
    rows, cols, ns = wl["rows"], wl["cols"], wl["streams"]
    total, POOL, ring = ns * world, 6, 4
    T, T0 = max(steps, gate_steps + 8), gate_steps + 2
    hp = make_hotpath(wl, dev.index, ring_depth=ring, n_streams=ns)      # (oatgpu_track_enqueue_dev: one frame a launch)
    pipe = FrameScatterPipe(total, (rows, cols, 3), dev, src=0, depth=depth, consumer=hp, via_host=(backend != "nccl"))
    pool = make_pool_device(rows, cols, total, POOL, 4242, dev) if rank == 0 else None

    def frames(t):
        return pool[0 if t == 0 else t % POOL] if rank == 0 else None

    def sync_all():
        hp.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
    kept, results = [], []
    try:
        pipe.post(0, frames(0))
        t_start = None
        for t in range(T):
            if t == T0:
                while hp.outstanding():
                    results.append(hp.collect())
                sync_all()
                t_start = time.perf_counter()
            if t + 1 < T:
                pipe.post(t + 1, frames(t + 1))
            local = pipe.take(t)
            if t <= gate_steps:
                kept.append(local.cpu().numpy().copy())          # what ARRIVED, for the oracle
            hp.enqueue_dev(local.data_ptr())
            if hp.outstanding() == ring:
                results.append(hp.collect())
        while hp.outstanding():
            results.append(hp.collect())
        sync_all()
        el = reduce_max([time.perf_counter() - t_start])[0]
        found = sum(1 for r_ in results[T0:] for q in r_ if q.position_valid)
        # ---- gate: the oracle from scratch over the frames this rank received ----
        verdict = "ok"
        p = oracle_params(wl)
        for s_ in range(ns):
            orc = oracle_mog(wl)
            for t in range(gate_steps + 1):
                want, _ = O.chain_step(orc, kept[t][s_], ALPHA, p, nthreads=host_threads())
                g = results[t][s_]
                if g.position_valid != want["valid"] or (want["valid"] and (
                        (g.a00, g.a10, g.a01) != (want["a00"], want["a10"], want["a01"]) or
                        abs(g.x - want["x"]) > 1e-4 or abs(g.y - want["y"]) > 1e-4)):
                    verdict = f"rank {rank} stream {s_} frame {t}: differs from the oracle"
                    break
            if verdict != "ok":
                break
        rec = dict(rank=rank, parity=verdict, found=found)
    except Exception as e:               # a broken transport must not take the per-rank-ingest line down with it
        el, rec = None, dict(rank=rank, parity=f"error: {str(e)[-160:]}", found=0)
        log(f"[rank {rank}] scatter leg failed:", e)
    finally:
        hp.close()
    allrec = [None] * world
    dist.all_gather_object(allrec, rec)
    bad = [q for q in allrec if q["parity"] != "ok"]
    out = dict(fps=(total * (T - T0) / el) if el else None, ms_per_step=(el / (T - T0) * 1e3) if el else None, steps=T - T0,
               depth=depth, backend=backend, bytes_per_peer=pipe.bytes_per_peer, streams_total=total,
               parity="ok" if not bad else bad[0]["parity"], per_rank=allrec,
               positions_found=sum(q["found"] for q in allrec), positions_expected=total * (T - T0),
               frames_per_launch=1,
               what="all frames originate on rank 0 and reach their ranks through FrameScatterPipe (one send/recv per peer and "
                    "step, double-buffered, the hot path as consumer); fresh models, one frame a launch; gate = the oracle from "
                    "scratch over the first frames every rank RECEIVED")
    pool = None
    torch.cuda.empty_cache()
    return out


def scatter_with_watchdog(args, world, rank, dev, reduce_max, line_so_far, t_start, also=None, extra_fn=None, want_scatter=True):
    """The OPTIONAL legs of an N > 1 run -- the second workload on every rank (extra_fn, round 6) and the scatter legs, the only
    part with a data-path exchange -- run LAST, behind the line's own measurements, under a timer: should a rank die in them or a
    transport hang (a peer that died, a P2P path that does not come up), the timer ends the rank instead of the job's default
    10-minute collective timeout killing it without a line -- rank 0 first prints the line it already has, marked.  The driver's
    multi-GPU run is a one-shot: `value` must not depend on legs that come after it."""
    import threading
    done = threading.Event()
    limit = args.scatter_timeout + (getattr(args, "extra_timeout", 240.0) if extra_fn else 0.0)

    def fire(why=None):
        if done.is_set():
            return
        done.set()
        why = why or f"no result within {limit:.0f} s"
        log(f"[rank {rank}] optional legs (second workload / scatter): {why} -- given up")
        if rank == 0 and line_so_far is not None:
            if want_scatter and not line_so_far.get("scatter_ingest"):
                line_so_far["scatter_ingest"] = dict(error=why, parity="timeout" if "within" in why else "error",
                                                     backend=args.backend)
            if extra_fn and not line_so_far.get("extra_workloads") and also:
                line_so_far["extra_workloads"] = {also: dict(name=also, value=None, parity="timeout" if "within" in why else "error")}
            line_so_far["bench_wall_s"] = time.perf_counter() - t_start
            emit(line_so_far)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    t = threading.Timer(limit, fire)
    t.daemon = True
    t.start()
    # (a peer that DIES in the leg makes the launcher terminate the others: rank 0 then still hands over the line it has)
    import signal
    old_term = None
    try:
        old_term = signal.signal(signal.SIGTERM, lambda *_: fire("terminated by the launcher (a peer died in the scatter leg)"))
    except ValueError:
        pass
    if extra_fn:                         # the other north-star size on every rank, gated (multi_rank_extra)
        try:
            ex = extra_fn()
        except Exception as e:
            log(f"[rank {rank}] second workload failed:", e)
            ex = dict(name=also, value=None, parity=f"error: {str(e)[-160:]}")
        if rank == 0 and line_so_far is not None:
            line_so_far["extra_workloads"] = {ex.get("name") or also: ex}
    if not want_scatter:
        done.set()
        t.cancel()
        if old_term is not None:
            signal.signal(signal.SIGTERM, old_term)
        return None
    try:
        sc = scatter_leg(args.workload, world, rank, dev, args.backend, args.scatter_steps, reduce_max)
    except Exception as e:
        log(f"[rank {rank}] scatter leg failed:", e)
        sc = dict(error=str(e)[-200:], parity="error", backend=args.backend)
    if rank == 0 and line_so_far is not None:
        line_so_far["scatter_ingest"] = sc          # (should the second scatter leg hang, the first one's result is in the line)
    if also:                             # the other workload's shard through the same scatter (fewer steps: it is the second leg)
        try:
            s2 = scatter_leg(also, world, rank, dev, args.backend, max(args.scatter_steps // 2, 12), reduce_max)
        except Exception as e:
            log(f"[rank {rank}] scatter leg ({also}) failed:", e)
            s2 = dict(error=str(e)[-200:], parity="error", backend=args.backend)
        sc["also"] = dict(s2, workload=also)
    done.set()
    t.cancel()
    if old_term is not None:
        signal.signal(signal.SIGTERM, old_term)
    return sc


# ------------------------------------------------------------------ PMC legs --

def pmc_child(args):
    """Child of a rocprofv3 --pmc pass: the plain K1-bearing loop, nothing else."""
    torch.cuda.set_device(0)
    leg = Leg(args.workload, 0, 0, dense=args.dense_model, pool=10 if args.dense_model else args.pool)
    # (the PMC passes run with --early-blob 0: rocprofv3 --pmc serialises kernel dispatches, and a blob workgroup dispatched
    # ahead of its row scan would wait for a kernel that cannot start -- oatgpu_set_early_blob.  --k1-wg pins the per-pixel
    # kernel to the workgroup size the BENCHED run used, so the counters are read on the very instantiation `value` ran)
    if args.k1_wg:
        leg.hp.set_k1_workgroup(args.k1_wg)
    leg.init()
    leg.age(args.age)
    leg.run(args.warmup)
    leg.run(args.steps)
    leg.hp.synchronize()
    leg.close()


def pmc_pass(workload, dense, counter, W, K, timeout_s=240, k1_wg=0):
    """rocprofv3 --kernel-trace --pmc <counter> around a child of this script (counters in their own pass, as
    MI355X_MICROARCH.md prescribes).  Returns (avg counter value in KiB per k_mog_fused dispatch over the last K
    dispatches, avg duration us, dispatches) or None."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="oat_pmc_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", tmp, "-o", "r", "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--pmc-child", "--workload", workload, "--steps", str(K), "--warmup", str(W),
           "--age", str(AGE if not dense else 60), "--mog-restore-nmodes", str(RESTORE), "--learning-rate", str(ALPHA),
           "--fusion", str(FUSION), "--early-blob", "0", "--k1-wg", str(k1_wg)]
    if dense:
        cmd.append("--dense-model")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
        db = None
        for dp, _, fs in os.walk(tmp):
            for f in fs:
                if f.endswith(".db"):
                    db = os.path.join(dp, f)
        if r.returncode != 0 or not db:
            log(f"pmc pass {workload} {counter}: rc={r.returncode} {r.stderr[-400:]}")
            return None
        con = sqlite3.connect(db)
        rows = con.execute("select kernel_name, value, duration from counters_collection where counter_name = ? "
                           "order by start", (counter,)).fetchall()
        # the last K dispatches of the per-pixel kernel, whichever instantiation each was (the library switches the
        # cache policy of its slot-1..4 loads with the model's density: two instantiations, one per step)
        lst = [(v, d) for k, v, d in rows if "k_mog_fused" in k][-K:]
        if not lst:
            return None
        val = sum(v for v, _ in lst) / len(lst)
        dur = sum(d for _, d in lst) / len(lst) / 1e3
        n = len(lst)
        return val, dur, n
    except Exception as e:      # never let a profiler problem break the benchmark line
        log(f"pmc pass {workload} {counter} failed: {e}")
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_traffic(workload, W, dense_audit_bytes, benched_is_dense, k1_wg_dense=0, k1_wg_benched=0):
    """HBM bytes per k_mog_fused launch of the dense leg and of the benched workload, as MI355X_MICROARCH.md
    prescribes: FETCH_SIZE and WRITE_SIZE in separate passes, KiB units, FETCH_SIZE doubled (on gfx950 it reports
    half the bytes of coalesced streaming reads).  `bytes_per_launch` = 2 x FETCH_SIZE + WRITE_SIZE.  The guide calls
    other access widths and WRITE_SIZE uncalibrated and asks for a calibration on a known byte count in the
    kernel's own access pattern: the dense pass is that -- the kernel's audit gives its read and written bytes
    exactly -- and `bytes_per_launch_calibrated` applies the two factors found there."""
    out = dict(method="rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE: separate child passes of this very run, "
                      "KiB units, averaged over the last 48 k_mog_fused dispatches (one pool cycle); bytes_per_launch = 2 x FETCH_SIZE + "
                      "WRITE_SIZE (gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md); *_calibrated = factors fitted "
                      "on the dense pass against the kernel's audited bytes")
    K = 48                      # one whole cycle of the 48-frame pool: the traffic varies with where the discs are
    df = pmc_pass("4k1", True, "FETCH_SIZE", 12, K, k1_wg=k1_wg_dense)
    dw = pmc_pass("4k1", True, "WRITE_SIZE", 12, K, k1_wg=k1_wg_dense)
    if not df or not dw:
        return None
    fr = fw = None
    if dense_audit_bytes:
        fr = dense_audit_bytes[0] / (df[0] * 1024.0)
        fw = dense_audit_bytes[1] / (dw[0] * 1024.0)
    out["calibration"] = dict(fetch_factor=fr, write_factor=fw,
                              audited_read_bytes=dense_audit_bytes[0] if dense_audit_bytes else None,
                              audited_written_bytes=dense_audit_bytes[1] if dense_audit_bytes else None)

    def entry(f, w, **kw):
        d = dict(FETCH_SIZE_KiB=f[0], WRITE_SIZE_KiB=w[0], dispatches=f[2], avg_duration_us=f[1],
                 bytes_per_launch=2.0 * f[0] * 1024 + w[0] * 1024, **kw)
        if fr and fw:
            d["bytes_per_launch_calibrated"] = fr * f[0] * 1024 + fw * w[0] * 1024
        return d
    out["dense"] = entry(df, dw, k1_workgroup=k1_wg_dense)
    if benched_is_dense:
        return out
    sf = pmc_pass(workload, False, "FETCH_SIZE", W, K, k1_wg=k1_wg_benched)
    sw = pmc_pass(workload, False, "WRITE_SIZE", W, K, k1_wg=k1_wg_benched)
    if sf and sw:
        out["benched"] = entry(sf, sw, workload=workload, warmup=W, k1_workgroup=k1_wg_benched)
    return out



# ------------------------------------------------------- drop-in process pipeline --

def _bin(name):
    return os.path.join(ROOT, "build", "bin", name)


def _serve_timed(consumers, serve_cmd, settle=3.0, timeout=240):
    """The reference's perf methodology (test/perf/framefilt-mog.sh:1-3, posidet-hsv.sh:1-3): start the consumer(s),
    let them attach, then `time` the frame server publishing its frames into shared memory.  Returns seconds."""
    procs = [subprocess.Popen(c, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE) for c in consumers]
    try:
        time.sleep(settle)
        t0 = time.perf_counter()
        r = subprocess.run(serve_cmd, capture_output=True, text=True, timeout=timeout)
        el = time.perf_counter() - t0
        if r.returncode != 0:
            raise RuntimeError(f"{serve_cmd[0]} rc={r.returncode}: {r.stderr[-300:]}")
        for p in procs:
            p.wait(timeout=60)
        return el, r.stdout
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


def pipeline_block(device):
    """BASELINE.md section 1 on this box, through the drop-in binaries over shared memory: 1000 frames of a 1000 x 1000
    BGR image through ONE consumer, wall clock of the frame server (`time oat frameserve test raw -f earth-1MP.jpg -c
    test.toml test`), for `framefilt mog` and `posidet hsv` with their default parameters, next to the reference's
    published figures; plus the latency of the fused tracker at 1080p from a frame being posted to shared memory
    to its position token arriving (tools: oat-latency-probe), free-running and camera-paced."""
    need = ["oat-frameserve-raw", "oat-framefilt-hip", "oat-posidet-hip", "oat-track-hip", "oat-latency-probe", "oat-clean-hip"]
    if not all(os.path.exists(_bin(b)) for b in need):
        return None
    import uuid
    from oat_amd.synth import SyntheticStream
    tag = "oat_b_" + uuid.uuid4().hex[:6]
    files, addrs = [], []

    def raw_file(rows, cols, nframes):
        st = SyntheticStream(rows, cols, 0, n_discs=2)
        path = f"/dev/shm/{tag}_{rows}x{cols}.raw"
        np.stack([st.frame(9 * t, with_discs=t > 0) for t in range(nframes)]).tofile(path)
        files.append(path)
        return path
    out = dict(methodology="wall clock of the frame server publishing 1000 frames of a 1000 x 1000 BGR image into shared "
                           "memory with ONE consumer attached (test/perf/framefilt-mog.sh:1-3, posidet-hsv.sh:1-3; the consumer "
                           "takes every frame: shm hand-off, H2D copy of the frame, kernels, result back over PCIe); default "
                           "parameters of the components, as the reference's runs",
               reference_published_fps={"framefilt mog (CUDA MOG, GTX 970)": 573, "framefilt mog (CPU MOG2, i7-5600U)": 75.7,
                                        "posidet hsv (CPU, i7-5820K)": 213, "posidet hsv (CPU, i7-5600U)": 135,
                                        "source": "test/perf/results.md:34-37,91-95,55-58,121-124 (other hardware)"})
    try:
        one = raw_file(1000, 1000, 1)
        dev = ["--gpu-index", str(device)]
        a, b = tag + "raw", tag + "flt"
        addrs += [a, b]
        el, _ = _serve_timed([[_bin("oat-framefilt-hip"), "mog", a, b] + dev],
                             [_bin("oat-frameserve-raw"), a, "-f", one, "--rows", "1000", "--cols", "1000", "-n", "1000"])
        out["framefilt_mog_1MP"] = dict(fps=1000 / el, real_s=el, vs_reference_cuda_mog=1000 / el / 573, vs_reference_cpu_mog2=1000 / el / 75.7)
        a, b = tag + "hsv", tag + "pos"
        addrs += [a, b]
        el, _ = _serve_timed([[_bin("oat-posidet-hip"), "hsv", a, b] + dev],
                             [_bin("oat-frameserve-raw"), a, "-f", one, "--rows", "1000", "--cols", "1000", "-C", "HSV", "-n", "1000"])
        out["posidet_hsv_1MP"] = dict(fps=1000 / el, real_s=el, vs_reference_cpu=1000 / el / 213)
        # latency of the whole fused chain at 1080p, frame posted -> position token received
        hd = raw_file(1080, 1920, 8)
        lat = {}
        for label, rate in (("free_running", None), ("paced_500fps", "500")):
            a, b = tag + "cam" + label[:1], tag + "trk" + label[:1]
            addrs += [a, b]
            probe = [_bin("oat-latency-probe"), a, b, "-f", hd, "--rows", "1080", "--cols", "1920", "-n", "1000"] + (["-r", rate] if rate else [])
            trk = [_bin("oat-track-hip"), a, b, "-a", "0.01", "--area", "[20,100000]", "-H", "[100,125]", "-S", "[150,256]",
                   "-V", "[100,256]", "-e", "3", "-d", "7", "--ring", "2"] + dev
            # the probe binds the frame node first; the tracker attaches to it and binds the position node; the probe serves then
            pp = subprocess.Popen(probe, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            time.sleep(0.5)
            tp = subprocess.Popen(trk, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            try:
                so, se = pp.communicate(timeout=60)
                tp.wait(timeout=60)
            finally:
                for p_ in (pp, tp):
                    if p_.poll() is None:
                        p_.kill()
            line = [l for l in so.splitlines() if l.startswith("{")]
            lat[label] = json.loads(line[-1]) if line else dict(error=(se or "")[-200:])
        out["track_1080p_latency"] = dict(
            what="oat-latency-probe -> oat-track-hip (mog + HSV + morphology + contour centroid, ring 2) -> probe: time from "
                 "sink.post() of the frame to the arrival of its position token, 1000 frames of 1920 x 1080 BGR in shared memory; "
                 "free_running = frames as fast as the tracker takes them (the tracker holds two in flight), paced = a 500 fps camera",
            **lat)
        # configs[3]'s per-GPU shard from the drop-in boundary: 8 free-running 1080p cameras -> ONE batched oat-track-hip
        # (camera-by-camera staging, ABI 6) -> 8 readers; aggregate rate incl. process start-up (tools/pipeline_fps.py)
        try:
            # (its own process group: on a timeout the frame servers, the tracker and the readers go with it)
            pr = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "pipeline_fps.py"), "--rows", "1080", "--cols", "1920",
                                   "--frames", "1500", "--fused", "--cameras", "8", "--ring", "4", "--timing"], stdout=subprocess.PIPE,
                                  stderr=subprocess.PIPE, text=True, start_new_session=True)
            try:
                so, se = pr.communicate(timeout=150)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(pr.pid, signal.SIGKILL)
                so, se = pr.communicate()
                se = "timeout; " + (se or "")
            r = argparse.Namespace(stdout=so or "", stderr=se or "")
            m = re.search(r"(\d+) tokens in ([0-9.]+) s = ([0-9.]+) fps aggregate", r.stdout)
            m2 = re.search(r"rounds 17\.\.: ([0-9.]+) fps aggregate", r.stdout)
            m3 = re.search(r"per round \(us\): (.*?); steady", r.stdout)
            out["track_8x1080p"] = (dict(fps_aggregate=float(m.group(3)), tokens=int(m.group(1)), real_s=float(m.group(2)),
                                         fps_steady=float(m2.group(1)) if m2 else None,
                                         tracker_loop_us_per_round=m3.group(1) if m3 else None,
                                         what="8 oat-frameserve-raw (free-running, 1500 frames each) -> one oat-track-hip with 8 SOURCEs "
                                              "and 8 SINKs, ring 4 -> 8 oat-posi-cout; fps_aggregate: wall clock from the start of the frame "
                                              "servers to the last token, process start-up included; fps_steady: the tracker's own clock "
                                              "from the end of its round 16 to the end of its last round (oat-track-hip --timing)")
                                    if m else dict(error=(r.stdout + r.stderr)[-200:]))
        except Exception as e:
            out["track_8x1080p"] = dict(error=str(e)[-200:])
    except Exception as e:
        out["error"] = str(e)[-300:]
    finally:
        subprocess.run([_bin("oat-clean-hip")] + addrs, capture_output=True)
        for f in files:
            try:
                os.unlink(f)
            except OSError:
                pass
    return out


# ------------------------------------------------------------------------ main --

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = devices of this node (default 1, or the launcher's WORLD_SIZE).  N > 1 without a launcher: "
                         "the script starts the N ranks itself through torch.distributed.run")
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="4k1", choices=sorted(WORKLOADS))
    ap.add_argument("--pool", type=int, default=48, help="distinct frame sets resident in HBM")
    ap.add_argument("--input", default="device", choices=["device", "host", "host-sync"],
                    help="device: frames resident in HBM (the headline); host / host-sync: PCIe-inclusive variants "
                         "(reported in DESIGN.md, never the headline)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to smoke-test the "
                         "multi-rank control flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--dense-model", action="store_true",
                    help="make the dense diagnostic the benched workload: input that keeps all 5 mixture modes live on "
                         "every pixel (K1 moves the full 205 B/px) and a threshold window nothing passes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 PMC child passes (roofline.traffic = null)")
    ap.add_argument("--no-dense-leg", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_workloads legs (1080p16, 1080p8, 1080p1, vga1)")
    ap.add_argument("--no-spin-up", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the drop-in process pipeline block (BASELINE.md section 1 methodology)")
    ap.add_argument("--quick", action="store_true", help="= --no-pmc --no-extra --no-cpu-baseline --no-pipeline (kernel A/B runs)")
    ap.add_argument("--check-steps", type=int, default=64,
                    help="timed steps of every stream replayed through the oracle after the run (0 = off)")
    ap.add_argument("--learning-rate", type=float, default=None,
                    help="MOG2 adaptation coefficient (default 0.01 = SURVEY 8d; 0 = Oat's default, frozen model)")
    ap.add_argument("--mog-restore-nmodes", type=int, default=1, choices=[0, 1],
                    help="1 (default): MOG2Invoker's `nmodes = nNewModes;` -- pruned modes keep their slot; 0: pruning "
                         "shrinks the mode count (round 1's reading); oracle/mog2.c 'Mode count'")
    ap.add_argument("--age", type=int, default=600,
                    help="frames every model has seen (untimed) before the W warm-up and the K timed steps: a model a few "
                         "frames old is a transient (SURVEY 8d measures behind a warm-up; with this reading of the mode "
                         "count the model keeps changing for a few hundred frames); 0 = as young as W makes it")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--fusion", type=int, default=2, choices=[1, 2],
                    help="frames per launch of the fused per-pixel kernel on the pipelined path (oatgpu_set_fusion): 2 = "
                         "two consecutive frames on one pass over the model (the library's default), 1 = one launch a frame")
    ap.add_argument("--early-blob", type=int, default=None, choices=[0, 1],
                    help="oatgpu_set_early_blob: 1 = the blob workgroup of a step is dispatched ahead of its row scan and waits "
                         "for it on the device; 0 = the plain launch order; default (no flag): the library's choice by shape "
                         "(oatgpu_set_early_blob(-1): on for at most three streams at 4 MP a step and more)")
    ap.add_argument("--k1-wg", type=int, default=0, choices=[0, 64, 256], help=argparse.SUPPRESS)   # PMC child: oatgpu_set_k1_workgroup
    ap.add_argument("--detail-out", default=None, help="where the full result goes (default: bench_detail.json beside bench.py)")
    ap.add_argument("--no-scatter", action="store_true", help="N > 1: skip the scatter_ingest leg (frames from rank 0 through "
                                                              "the stream->rank scatter)")
    ap.add_argument("--scatter-steps", type=int, default=200, help="steps of the scatter_ingest leg (N > 1)")
    ap.add_argument("--scatter-timeout", type=float, default=150.0, help="seconds after which a hanging scatter leg is given up")
    ap.add_argument("--extra-timeout", type=float, default=240.0, help="N > 1: seconds the second workload's leg adds to that limit")
    ap.add_argument("--dense-noise", type=int, default=5, help=argparse.SUPPRESS)   # lab: 3 keeps every dense pixel background (no shadow test)
    ap.add_argument("--lab-calm", action="store_true", help=argparse.SUPPRESS)   # kernel lab: SURVEY 8d input WITHOUT the flickering pixels
    args = ap.parse_args()
    world = 1 if args.pmc_child else spawn_ranks_if_asked(args, sys.argv[1:])
    global DETAIL_PATH
    if args.detail_out:
        DETAIL_PATH = os.path.abspath(args.detail_out)
    global ALPHA, RESTORE, AGE, LAB_CALM, FUSION, DENSE_NOISE, EARLY_BLOB
    EARLY_BLOB = None if args.early_blob is None else bool(args.early_blob)
    DENSE_NOISE = args.dense_noise
    FUSION = args.fusion
    AGE = args.age
    LAB_CALM = args.lab_calm
    RESTORE = args.mog_restore_nmodes
    if args.learning_rate is not None:
        ALPHA = args.learning_rate
    if args.pmc_child:
        return pmc_child(args)
    if args.quick:
        args.no_pmc = args.no_extra = args.no_cpu_baseline = args.no_pipeline = True

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        ndev = torch.cuda.device_count()
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            local_rank = local_rank % max(ndev, 1)          # ranks may share a GPU in the smoke test
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback for the product path)"
    dev = torch.device("cuda", local_rank)
    open_retries = open_device_with_retry(dev, rank)
    red_dev = dev if (world == 1 or args.backend == "nccl") else torch.device("cpu")
    solo = world == 1

    wl = WORKLOADS[args.workload]
    rows, cols, ns = wl["rows"], wl["cols"], wl["streams"]
    K, W = args.steps, args.warmup
    t_start = time.perf_counter()
    numa_node = pin_to_gpu_node(local_rank)

    leg = Leg(args.workload, local_rank, rank, dense=args.dense_model, pool=10 if args.dense_model else args.pool,
              input_mode=args.input)
    log(f"[rank {rank}] device open, {args.pool}-frame pool of {args.workload} resident")

    def barrier():
        leg.hp.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def reduce_max(vals):
        if world == 1:
            return list(vals)
        t = torch.tensor(list(vals), dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t.tolist()]

    def local_barrier(l):
        return lambda: (l.hp.synchronize(), torch.cuda.synchronize())

    # Both parity gates run on EVERY rank for every stream it owns (r04): a rank whose shard differs from the oracle
    # must not hide behind rank 0's "ok"; the verdicts are gathered below and the line carries the worst of them.
    want_gate = not args.no_parity and not args.dense_model and args.input == "device"
    global THREAD_SHARE
    THREAD_SHARE = world
    # stream s lives on rank s // ceil(S / N) for life (oat_amd.dist.stream_partition); with the weak-scaling
    # workloads every rank owns `ns` streams, rank r the global streams r*ns .. r*ns + ns - 1
    from oat_amd.dist import stream_partition
    mine = stream_partition(ns * world, world, rank)
    assert len(mine) == ns and mine.start == rank * ns, (list(mine), rank, ns, world)
    tr = timed_run(leg, K, W, barrier, prof_every=8 if K >= 64 else 1, age_frames=AGE, export=want_gate,
                   spin=0.0 if args.no_spin_up else 0.35, spin_args=(args.workload, local_rank, rank),
                   reduce_max=reduce_max, isolated=True)
    positions, prof, models, handover, aged = tr["positions"], tr["prof"], tr["models"], tr["handover"], tr["aged"]
    n_found_local = tr["found"]

    # (the parity gates -- minutes of many-threaded CPU work -- run after ALL device timing of this process: the small
    # workloads are bound by the host's launch calls and measured up to 40 % lower behind them)

    # the kernel's own traffic count and the model's mode histogram, continuing this run's model
    aud = hist = None
    hbm_read = hbm_copy = None
    if rank == 0 and args.input == "device":
        try:
            aud = audit(leg)
            hist = mode_histogram(leg)
            hbm_read, hbm_copy = leg.hp.measure_hbm(1 << 30, 5)
        except Exception as e:          # never let a probe break the benchmark line
            log("audit / histogram / hbm probe failed:", e)

    if world > 1:
        found = torch.tensor([n_found_local], dtype=torch.int64, device=red_dev)
        dist.all_reduce(found, op=dist.ReduceOp.SUM)
        n_found = int(found.item())
    else:
        n_found = n_found_local

    log(f"[rank {rank}] timed run and probes done")
    k1_wg_benched, early_benched = leg.hp.last_step_shape()
    early_timeouts = leg.hp.early_blob_timeouts()
    one_lat = None
    if rank == 0 and args.input == "device":
        try:
            one_lat = single_frame_latency(leg, 300)
        except Exception as e:
            log("single-frame latency probe failed:", e)
    want_scatter = world > 1 and not args.no_scatter and args.input == "device" and not args.dense_model

    per_rank = None
    if world > 1:
        # N > 1: every rank gates its own shard now (nothing else is timed in this process afterwards), then the
        # verdicts, the partition and the per-rank counts travel to rank 0
        my_par, my_detail = "skipped", None
        if want_gate:
            my_par, my_detail = gates(leg, tr["gate_offset"], K, positions, args.check_steps, models, handover)
        from oat_amd import ffi as _ffi
        mine_rec = dict(rank=rank, device=local_rank, streams=[mine.start, mine.stop], parity=my_par,
                        device_open_retries=open_retries + _ffi.load().oatgpu_device_open_retries(),
                        positions_found=n_found_local, block_ms=tr["block_s"] * 1e3,
                        k_mog_fused_ms=k1_ms(prof)[0], ms_per_step_local=tr["local_block_s"] / K * 1e3)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_rec)

    # N > 1: the OTHER north-star size, every rank, gated (one invocation = the whole multi-GPU record) -- run with the scatter
    # legs, LAST and under their watchdog: nothing `value` needs comes after it
    extra_fn = None
    other = "1080p8" if args.workload != "1080p8" else "4k1"
    if world > 1 and not args.no_extra and args.input == "device" and not args.dense_model:
        extra_fn = lambda: multi_rank_extra(other, K, W, local_rank, rank, world, reduce_max, args)

    if rank != 0:
        if world > 1:
            if want_scatter or extra_fn:
                leg.close()
                scatter_with_watchdog(args, world, rank, dev, reduce_max, None, t_start, also=other if extra_fn else None,
                                      extra_fn=extra_fn, want_scatter=want_scatter)
            dist.barrier()              # rank 0 prints before everybody leaves
            dist.destroy_process_group()
        return

    # ---- the other BASELINE configs: device timing now, their gates later ----
    extra_runs = []
    if solo and not args.no_extra and args.input == "device" and not args.dense_model:
        # (configs[2], configs[3]'s per-GPU shard, configs[1], configs[0]'s shape)
        for name, kk, ww in (("1080p16", 100, 40), ("1080p8", 150, 40), ("1080p1", 500, 100), ("vga1", 1000, 100)):
            if name == args.workload:
                continue
            try:
                el_ = Leg(name, local_rank, rank, pool=24 if name == "1080p16" else 48)
                er = timed_run(el_, kk, ww, local_barrier(el_), 8, age_frames=AGE, export=not args.no_parity,
                               spin=0.0 if args.no_spin_up else 0.2, spin_args=(name, local_rank, rank))
                er.update(name=name, leg=el_, K=kk, W=ww, aud=audit(el_, 4))
                extra_runs.append(er)
            except Exception as e:
                log(f"extra workload {name} failed:", e)

    # ---- the benched workload once more with ONE frame a launch (oatgpu_set_fusion(1)): what a caller gets that
    # collects every frame before it hands over the next (SURVEY 8b read literally: nothing batched across time) ----
    one_frame = None
    if solo and FUSION == 2 and not args.no_extra and args.input == "device" and not args.dense_model:
        try:
            l1 = Leg(args.workload, local_rank, rank, pool=args.pool)
            l1.hp.set_fusion(1)
            o = timed_run(l1, K, max(W, 50), local_barrier(l1), 8, age_frames=AGE, export=False,
                          spin=0.0 if args.no_spin_up else 0.2, spin_args=(args.workload, local_rank, rank))
            one_frame = dict(value=ns * K / o["block_s"], unit="frames/s", steps=K, blocks=o["n_blocks"],
                             ms_per_step=o["block_s"] / K * 1e3, k_mog_fused_ms=k1_ms(o["prof"])[0],
                             frames_per_launch=o["prof"]["mog_frames"] / max(o["prof"]["steps"], 1),
                             note="same workload, same ageing, same block timing, oatgpu_set_fusion(1): one launch of the "
                                  "per-pixel kernel per frame; not gated in this run (the -m gpu tests compare both forms "
                                  "with the oracle and with each other)")
            l1.close()
            del l1
            torch.cuda.empty_cache()
        except Exception as e:
            log("one-frame-a-launch leg failed:", e)

    # ---- the benched workload at Oat's DEFAULT learning rate, `framefilt mog` without -a (BackgroundSubtractorMOG.cpp:51-67,
    # adaptation_coeff 0; SURVEY 8d: "and a second run at alpha = 0 = Oat default"): frame 1 learns at 1/2 (OpenCV's automatic
    # rate), every later frame leaves the model as it is -- one mode a pixel, read-only: 3 + 1 + 20 B/px a frame ----
    frozen = None
    if solo and not args.no_extra and args.input == "device" and not args.dense_model and ALPHA != 0.0:
        keep_alpha = ALPHA
        try:
            ALPHA = 0.0
            l0 = Leg(args.workload, local_rank, rank, pool=args.pool)
            o = timed_run(l0, K, max(W, 50), local_barrier(l0), 8, age_frames=60, export=not args.no_parity,
                          spin=0.0 if args.no_spin_up else 0.2, spin_args=(args.workload, local_rank, rank))
            f_k1 = k1_ms(o["prof"])[0]
            f_fpl = o["prof"]["mog_frames"] / max(o["prof"]["steps"], 1)
            f_aud = audit(l0, 4)
            f_par = "skipped"
            if not args.no_parity:
                f_par, _ = gates(l0, o["gate_offset"], K, o["positions"], min(args.check_steps, 16), o["models"], o["handover"])
            frozen = dict(value=ns * K / o["block_s"], unit="frames/s", learning_rate=0.0, steps=K, blocks=o["n_blocks"],
                          ms_per_step=o["block_s"] / K * 1e3, k_mog_fused_ms=f_k1, frames_per_launch=f_fpl,
                          useful_bytes_per_px=f_aud["useful_read_B_per_px"] + f_aud["useful_write_B_per_px"],
                          audit_frames_per_launch=f_aud["frames_per_launch"], parity=f_par,
                          note="Oat's default: oat framefilt mog without -a (adaptation_coeff 0): the first frame initialises "
                               "the model, later frames only classify against it (one mode a pixel, nothing stored); same "
                               "workload, same block timing, both parity gates at this rate")
            l0.close()
            del l0
            torch.cuda.empty_cache()
        except Exception as e:
            log("default-learning-rate leg failed:", e)
        finally:
            ALPHA = keep_alpha

    total_streams = ns * world
    block_s = tr["block_s"]
    fps = total_streams * K / block_s
    px_per_launch = rows * cols * ns
    mog_ms, mog_ms_raw = k1_ms(prof)
    fpl = prof["mog_frames"] / max(prof["steps"], 1)          # frames a launch of the per-pixel kernel covered (1..2)
    pool_host0 = [leg.pool[leg.pool_index(i)][0].cpu().numpy() for i in range(8)]
    benched = dict(workload=args.workload + (" --dense-model" if args.dense_model else ""), avg_launch_ms=mog_ms,
                   avg_launch_ms_raw_events=mog_ms_raw, empty_event_pair_ms=prof["event_pair_ms"],
                   px_per_launch=px_per_launch, frames_per_launch=fpl, k_mog_fused_ms_per_frame=mog_ms / fpl,
                   note="a launch takes frames_per_launch consecutive frames of every stream on ONE pass over the model "
                        "(oatgpu_set_fusion); avg_launch_ms, traffic and moved_bytes_per_px are per LAUNCH, "
                        "moved_bytes_per_px_frame = that / frames_per_launch; useful_* / requested_* are the kernel's own "
                        "audit of its launches (audit_frames_per_launch frames each)")
    if aud:
        # the audit counts the launches as the library makes them (two frames a launch on the pipelined path): per pixel
        # and LAUNCH.  Where its launches and the timed ones differ in frames (GREY contexts audit one frame at a time)
        # the second frame's 3 B/px and threshold words are added as a lower bound.
        u1 = aud["useful_read_B_per_px"] + aud["useful_write_B_per_px"]
        a_fpl = aud["frames_per_launch"]
        benched.update(useful_bytes_per_px=u1, audit_frames_per_launch=a_fpl,
                       useful_bytes_per_px_frame=u1 / max(a_fpl, 1e-9),
                       requested_sector32_bytes_per_px=aud["sector32_read_B_per_px"] + aud["sector32_write_B_per_px"],
                       requested_sector64_bytes_per_px=aud["sector64_read_B_per_px"] + aud["sector64_write_B_per_px"],
                       useful_bytes_per_px_launch_lower_bound=u1 + max(fpl - a_fpl, 0.0) * (FRAME_BYTES_PER_PIXEL + 0.125),
                       audit=aud, mode_histogram=hist)
    # ---- the leg where the algorithmic bytes really move: 4K, all five modes live on every pixel ----
    dense = None
    k1_wg_dense = k1_wg_benched
    if solo and not args.no_dense_leg and args.input == "device":
        try:
            if args.dense_model and args.workload == "4k1":
                dense = dict(avg_launch_ms=mog_ms, px_per_launch=px_per_launch, steps=K, audit=aud, frames_per_launch=fpl)
            else:
                dl = Leg("4k1", local_rank, rank, dense=True, pool=10)
                # 1 200 untimed warm-up steps (~0.35 s): a device that has just become busy runs this leg 10-15 %
                # faster for its first ~100 ms than it sustains (r02: 235 us in a 32-ms window against 272 us in the
                # PMC child pass of the same run) -- the fraction on the line is the SUSTAINED one
                torch.cuda.synchronize()
                time.sleep(0.25)             # let the device fall idle: the burst window below starts from rest
                b_prof = timed_run(dl, 100, 20, local_barrier(dl), 1, age_frames=60, export=False, min_ms=0.0)["prof"]
                dr = timed_run(dl, 300, 1200, local_barrier(dl), 2, age_frames=60, export=False, min_ms=0.0)
                d_prof = dr["prof"]
                k1_wg_dense = dl.hp.last_step_shape()[0]
                d_aud = audit(dl, 4)
                dense = dict(avg_launch_ms=k1_ms(d_prof)[0], px_per_launch=3840 * 2160, steps=300,
                             ms_per_step=dr["block_s"] / 300 * 1e3, audit=d_aud, burst_avg_launch_ms=k1_ms(b_prof)[0],
                             frames_per_launch=d_prof["mog_frames"] / max(d_prof["steps"], 1))
                dl.close()
                del dl
                torch.cuda.empty_cache()
                # the same leg on an input whose pixels all STAY background (noise +-3 instead of +-5: no lane of any wave
                # creates a mode, converts a foreground pixel to HSV or walks past its own mode): the bytes without the extra arithmetic
                try:
                    keep_noise, DENSE_NOISE = DENSE_NOISE, 3
                    dq = Leg("4k1", local_rank, rank, dense=True, pool=10)
                    DENSE_NOISE = keep_noise
                    rq = timed_run(dq, 300, 1200, local_barrier(dq), 2, age_frames=60, export=False, min_ms=0.0)
                    q_aud = audit(dq, 4)
                    dense["quiet"] = dict(avg_launch_ms=k1_ms(rq["prof"])[0], ms_per_step=rq["block_s"] / 300 * 1e3,
                                          frames_per_launch=rq["prof"]["mog_frames"] / max(rq["prof"]["steps"], 1), audit=q_aud)
                    dq.close()
                    del dq
                    torch.cuda.empty_cache()
                except Exception as e:
                    DENSE_NOISE = keep_noise
                    log("quiet dense leg failed:", e)
                if dense["frames_per_launch"] > 1.5:
                    # the same leg with one frame a launch (oatgpu_set_fusion(1): what a caller that collects every
                    # frame before the next gets, and SURVEY 8d's 205 B/px case), sustained state as above
                    d1 = Leg("4k1", local_rank, rank, dense=True, pool=10)
                    d1.hp.set_fusion(1)
                    r1 = timed_run(d1, 300, 1200, local_barrier(d1), 2, age_frames=60, export=False, min_ms=0.0)
                    dense["one_frame_avg_launch_ms"] = k1_ms(r1["prof"])[0]
                    dense["one_frame_ms_per_step"] = r1["block_s"] / 300 * 1e3
                    d1.close()
                    del d1
                    torch.cuda.empty_cache()
        except Exception as e:
            log("dense leg failed:", e)

    # ---- parity gates of everything timed above (CPU), then the legs can go ----
    parity, parity_detail = "skipped", None
    if want_gate and world > 1:
        bad = [r for r in per_rank if r["parity"] != "ok"]
        parity = "ok" if not bad else f"rank {bad[0]['rank']}: {bad[0]['parity']}"
        parity_detail = (f"both gates on every one of the {world} ranks, every stream of its shard (fresh-context masks + "
                         f"centroids on 4 frames; first {min(args.check_steps, K)} timed steps vs the oracle continuing from "
                         f"the device's exported model): " + ", ".join(f"rank {r['rank']} {r['parity']}" for r in per_rank))
    elif want_gate:
        parity, parity_detail = gates(leg, tr["gate_offset"], K, positions, args.check_steps, models, handover)
    models = None
    leg.close()
    del leg
    extra = {}
    for er in extra_runs:
        e_par = "skipped"
        if not args.no_parity:
            e_par, _ = gates(er["leg"], er["gate_offset"], er["K"], er["positions"], min(args.check_steps, 16), er["models"], er["handover"])
        w_ = WORKLOADS[er["name"]]
        ppl = w_["rows"] * w_["cols"] * w_["streams"]
        e_k1 = k1_ms(er["prof"])[0]
        e_fpl = er["prof"]["mog_frames"] / max(er["prof"]["steps"], 1)
        extra[er["name"]] = dict(value=w_["streams"] * er["K"] / er["block_s"], unit="frames/s", steps=er["K"], blocks=er["n_blocks"],
                                 warmup=er["W"], model_age_frames=er["handover"], ms_per_step=er["block_s"] / er["K"] * 1e3,
                                 k_mog_fused_ms=e_k1, frames_per_launch=e_fpl, k_mog_fused_ms_per_frame=e_k1 / e_fpl, px_per_launch=ppl,
                                 useful_bytes_per_px=er["aud"]["useful_read_B_per_px"] + er["aud"]["useful_write_B_per_px"],
                                 requested_sector32_bytes_per_px=er["aud"]["sector32_read_B_per_px"] + er["aud"]["sector32_write_B_per_px"],
                                 audit_frames_per_launch=er["aud"]["frames_per_launch"], parity=e_par)
        er["leg"].close()
        er["models"] = er["leg"] = None
    extra_runs = []
    torch.cuda.empty_cache()

    pmc = None
    if solo and dense and not args.no_pmc and args.input == "device":
        t0 = time.perf_counter()
        da = dense.get("audit")
        # calibration of the two counters against the kernel's own byte count: only where the audited launches and the
        # profiled ones are the same kind (the audit counts one-frame launches)
        aud_bytes = ((da["sector32_read_B_per_px"] * dense["px_per_launch"],
                      da["sector32_write_B_per_px"] * dense["px_per_launch"])
                     if da and abs((dense.get("frames_per_launch") or 1.0) - da.get("frames_per_launch", 1.0)) < 1e-6 else None)
        pmc = pmc_traffic(args.workload, W, aud_bytes, args.dense_model, k1_wg_dense=k1_wg_dense, k1_wg_benched=k1_wg_benched)
        log(f"pmc passes: {time.perf_counter() - t0:.1f} s")

    roofline = {"bound": "hbm", "kernel": "k_mog_fused", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "measured_stream_read_GBps": hbm_read, "measured_stream_copy_GBps": hbm_copy}
    if dense:
        # What a launch has to move, as executed: frames_per_launch frames in (3 B/px each), the model in and out ONCE
        # (101 + 101 B/px).  With one frame a launch that is SURVEY 8d's 205 B/px; with two frames a launch 208 B/px
        # for two frames -- the kernel keeps the mixture in registers between the frames, which is the point of it.
        d_fpl = dense.get("frames_per_launch") or 1.0
        launch_bytes = (MODEL_BYTES_PER_PIXEL + FRAME_BYTES_PER_PIXEL * d_fpl) * dense["px_per_launch"]
        a_dense = launch_bytes / (dense["avg_launch_ms"] * 1e-3) / 1e9
        d_tr = (pmc or {}).get("dense", {}).get("bytes_per_launch")
        da = dense.get("audit") or {}
        d_req = (da.get("sector32_read_B_per_px", 0) + da.get("sector32_write_B_per_px", 0)) * dense["px_per_launch"]
        if abs(d_fpl - da.get("frames_per_launch", 1.0)) > 1e-6:
            d_req = 0                    # the audit describes one-frame launches, this leg ran two frames a launch
        per_frame_equiv = BYTES_PER_PIXEL * d_fpl * dense["px_per_launch"] / (dense["avg_launch_ms"] * 1e-3) / 1e9
        roofline.update(
            leg="4k1 --dense-model, run inside this process: every pixel keeps five live modes and never matches the "
                "first one, so every lane loads the whole model; stores go out for planes whose bits changed "
                "(dense_audit); HIP events on the kernel's own stream",
            achieved=a_dense, frac=a_dense / HBM_PEAK_GBPS, frac_dense=a_dense / HBM_PEAK_GBPS,
            frames_per_launch=d_fpl, bytes_per_launch=launch_bytes, avg_launch_ms=dense["avg_launch_ms"],
            avg_ms_per_frame=dense["avg_launch_ms"] / d_fpl,
            one_pass_per_frame_equivalent_GBps=per_frame_equiv,
            dense_audit=dense.get("audit"), traffic=d_tr,
            frac_dense_traffic=(d_tr / (dense["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS) if d_tr else None,
            frac_dense_requested=(d_req / (dense["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS) if d_req else None,
            note="achieved = the bytes a launch must move AS EXECUTED -- (202 + 3 x frames_per_launch) B/px: the model in "
                 "and out once, frames_per_launch frames in; 205 B/px with one frame a launch (SURVEY 8d), 208 B/px for "
                 "TWO frames with two (--fusion 2, the library's default: the mixture stays in registers between two "
                 "consecutive frames) -- / kernel time on the leg built to move them, in the SUSTAINED state (300 steps "
                 "timed behind 1 200 warm-up steps); frac <= 1 by construction.  one_pass_per_frame_equivalent_GBps = "
                 "205 B/px x frames_per_launch / kernel time: what a kernel that re-reads the model for every frame "
                 "would have to sustain for the same frame rate -- above the pin rate with two frames a launch, NOT a "
                 "roofline fraction.  frac_dense_traffic = the launch priced at its PMC bytes, frac_dense_requested at "
                 "the 32-byte sectors the kernel itself counted; *_burst = the first 100 steps of a device that was idle")
        if dense.get("one_frame_avg_launch_ms"):
            t1 = dense["one_frame_avg_launch_ms"]
            roofline["one_frame_a_launch"] = dict(
                avg_launch_ms=t1, bytes_per_launch=BYTES_PER_PIXEL * dense["px_per_launch"],
                achieved=BYTES_PER_PIXEL * dense["px_per_launch"] / (t1 * 1e-3) / 1e9,
                frac=BYTES_PER_PIXEL * dense["px_per_launch"] / (t1 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                note="the same dense leg with oatgpu_set_fusion(1): SURVEY 8d's 205 B/px per launch, sustained state")
            roofline["frac_one_frame"] = roofline["one_frame_a_launch"]["frac"]
            roofline["value_dense_fps_one_frame"] = (1e3 / dense["one_frame_ms_per_step"]) if dense.get("one_frame_ms_per_step") else None
        if dense.get("burst_avg_launch_ms"):
            roofline.update(avg_launch_ms_burst=dense["burst_avg_launch_ms"],
                            frac_burst=launch_bytes / (dense["burst_avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS)
        if dense.get("quiet"):
            q = dense["quiet"]
            qb = (MODEL_BYTES_PER_PIXEL + FRAME_BYTES_PER_PIXEL * q["frames_per_launch"]) * dense["px_per_launch"]
            qa = q.get("audit") or {}
            roofline["all_background"] = dict(
                avg_launch_ms=q["avg_launch_ms"], bytes_per_launch=qb, achieved=qb / (q["avg_launch_ms"] * 1e-3) / 1e9,
                frac=qb / (q["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, value_dense_fps=1e3 / q["ms_per_step"],
                audited_sector_bytes_per_px=qa.get("sector32_read_B_per_px", 0) + qa.get("sector32_write_B_per_px", 0),
                note="the dense leg with per-frame noise +-3 instead of +-5: every pixel still cycles through its five modes and "
                     "every lane loads and stores the whole model (audited bytes beside it), but no pixel leaves the background, "
                     "so no wave runs the no-fit paths.  In the +-5 leg that `frac` is quoted on, one or two lanes of most waves "
                     "leave the background: their waves create a mode, walk all five and convert the pixel to HSV (1 005 against 562 "
                     "vector instructions a wave, profiles/r03p_k1_sq_counters_dense.md; the shadow test, which nothing on the fused "
                     "path can observe, is not evaluated since r03 -- DESIGN.md section 3)")
            roofline["frac_all_background"] = roofline["all_background"]["frac"]
        # the frame rate that belongs next to `frac`: whole chain on the dense model (one 4K stream)
        roofline["value_dense_fps"] = 1e3 / dense["ms_per_step"] if dense.get("ms_per_step") else None
    else:
        roofline.update(leg=None, achieved=None, frac=None, traffic=None,
                        note="dense leg not run (N > 1, --no-dense-leg or host input): no defensible fraction on this line")
    if pmc and pmc.get("benched"):
        b = pmc["benched"]["bytes_per_launch"]
        benched.update(traffic=b, moved_bytes_per_px=b / px_per_launch, moved_bytes_per_px_frame=b / px_per_launch / fpl,
                       frac_real=b / (mog_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS)
        if aud:                          # moved per launch / what the launch must ask for at least
            benched["waste_ratio"] = benched["moved_bytes_per_px"] / max(benched["useful_bytes_per_px_launch_lower_bound"], 1e-9)
    roofline["frac_real"] = benched.get("frac_real")
    roofline["frac_benched"] = benched.get("frac_real")      # the workload `value` is measured on: PMC bytes / kernel time / peak
    roofline["frac_benched_source"] = "pmc" if benched.get("frac_real") is not None else None
    if world > 1 and per_rank:
        # N > 1 (no profiler child passes): the SLOWEST rank's per-pixel launch, by HIP events on its own stream, priced at the
        # bytes the kernel itself counted on rank 0's model (32-byte sectors its loads and stores touch; the ranks run the same
        # synthetic input on models of the same age) -- the fraction a SCALE line carries
        k1s = [q["k_mog_fused_ms"] for q in per_rank if q.get("k_mog_fused_ms")]
        if k1s:
            benched.update(avg_launch_ms=max(k1s), avg_launch_ms_min_rank=min(k1s), avg_launch_ms_rank0=mog_ms)
            roofline["k_mog_fused_ms_ranks"] = {"min": min(k1s), "max": max(k1s)}
            req = benched.get("requested_sector32_bytes_per_px")
            if req and aud and abs(aud["frames_per_launch"] - fpl) < 1e-6:
                b = req * px_per_launch
                benched.update(traffic=b, traffic_source="kernel audit (32-byte sectors), rank 0")
                roofline["frac_benched"] = b / (max(k1s) * 1e-3) / 1e9 / HBM_PEAK_GBPS
                roofline["frac_benched_source"] = "audit"
    naive = BYTES_PER_PIXEL * rows * cols * ns / (block_s / K) / 1e9      # SURVEY 8d's per-frame figure x the benched frame rate
    roofline["algorithmic_205B_x_fps_GBps"] = naive
    roofline["fractions"] = (
        "frac = the DENSE leg (all five modes live, every lane loads the whole model) at the bytes a launch must move as "
        "executed, (202 + 3 x frames_per_launch) B/px, over its HIP-event time: the kernel against the memory system.  "
        "frac_one_frame = the same leg with one frame a launch: SURVEY 8d's 205 B/px per launch, verbatim.  frac_benched = "
        "the workload `value` is measured on (SURVEY 8d's synthetic input on aged models), priced at the HBM bytes the PMC "
        f"counters saw per launch.  205 B/px x the benched frame rate = {naive:.0f} GB/s "
        + ("EXCEEDS the 8 000 GB/s peak" if naive > HBM_PEAK_GBPS else "is below the peak")
        + ": the benched launch does not move the algorithmic bytes -- the everyday model is sparse (mean live modes in "
        "benched_workload.mode_histogram; dead slots are neither read nor written) and, with two frames a launch, the "
        "model crosses HBM once per two frames.  value_dense_fps is the frame rate that belongs next to frac.")
    roofline["useful_bytes_per_px"] = benched.get("useful_bytes_per_px")
    roofline["waste_ratio"] = benched.get("waste_ratio")
    roofline["benched_workload"] = benched
    roofline["pmc"] = pmc

    region_ms = tr["region_s"] * 1e3
    line = {
        "metric": "frames/sec/GPU (1080p & 4K) mog+hsv+ccl fused; % HBM roofline",
        "value": fps,
        # the whole timed region / its steps (every block, the first one's pipeline fill included): `value` is the MEDIAN block's
        "value_mean": total_streams * tr["steps_timed"] / tr["region_s"],
        "value_one_frame_a_launch": one_frame["value"] if one_frame else None,
        "value_default_learning_rate_0": frozen["value"] if frozen else None,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": block_s / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        # (SURVEY 8d's input puts ONE flickering pixel into every 64-pixel wave: every wave of the per-pixel kernel walks its
        # no-fit paths -- the worst case for that kernel by construction, and the contract's input: VERDICT r04 weak-5)
        "data": ("synthetic (SURVEY 8d; one flickering px in every 64-px wave: K1's worst case)" if not LAB_CALM
                 else "synthetic, LAB: no flickering pixels (not the SURVEY 8d input)"),
        "config": {"workload": f"{ns} x {cols}x{rows} uchar3 stream(s) per GPU, MOG2(5 mixtures, lr {ALPHA}) + HSV + "
                               f"inRange + erode {wl['erode']} + dilate {wl['dilate']} + external-contour centroid",
                   "name": args.workload, "streams_per_gpu": ns, "rows": rows, "cols": cols,
                   "erode": wl["erode"], "dilate": wl["dilate"],
                   "learning_rate": ALPHA, "mog_restore_nmodes": RESTORE, "model_age_frames": handover,
                   "frames_per_launch": fpl, "k1_workgroup": k1_wg_benched, "early_blob": early_benched,
                   "parallelism": f"streams sharded, {world} rank(s)"},
        "fps_per_gpu": fps / world,
        "partition": {"rule": "stream s -> rank s // ceil(S / N), contiguous blocks, for life (SURVEY 8e; oat_amd.dist.stream_partition)",
                      "streams_total": total_streams,
                      "per_rank": per_rank if per_rank else [dict(rank=0, device=local_rank, streams=[0, ns], parity=parity,
                                                                  positions_found=n_found_local, k_mog_fused_ms=mog_ms)]},
        "timing": {
            "method": f"{tr['n_blocks']} back-to-back blocks of exactly --steps {K} steps in ONE pipelined run between two "
                      "barriers (+ device synchronisation), sized so that the region lasts >= "
                      f"{MIN_TIMED_MS:.0f} ms whatever --steps is; a block's time runs from the collection of the previous "
                      "block's last result to the collection of its own last result (max over ranks per block); "
                      "ms_per_step = MEDIAN of blocks 1.. / steps (block 0 also carries the pipeline's fill).  "
                      "isolated_block = the same K steps ALONE between two barriers, fill and drain inside: what "
                      "rounds 1-2 put into `value`",
            "blocks": tr["n_blocks"], "steps_timed": tr["steps_timed"], "timed_region_ms": region_ms,
            "whole_region_ms_per_step": region_ms / tr["steps_timed"],
            "block_ms": {"first": tr["blocks"][0] * 1e3, "median": block_s * 1e3,
                         "min": min(tr["blocks"][1:] or tr["blocks"]) * 1e3, "max": max(tr["blocks"][1:] or tr["blocks"]) * 1e3},
            "isolated_block_ms": tr["isolated_block_s"] * 1e3 if tr["isolated_block_s"] else None,
            "value_isolated_block": (total_streams * K / tr["isolated_block_s"]) if tr["isolated_block_s"] else None,
            "calibration_ms_per_step": tr["step_s_calibration"] * 1e3,
        },
        "timed_region_ms": region_ms,
        "model_age_frames": handover,
        "host_numa_node": numa_node,
        "roofline": roofline,
        "stage_ms": {"mog": mog_ms, "morph": prof["morph_ms"] / max(prof["steps"], 1),
                     "blob": prof["blob_ms"] / max(prof["steps"], 1),
                     "gpu_total": prof["total_ms"] / max(prof["steps"], 1)},
        # time from a frame's hand-over to its result, (a) in the saturated run `value` is measured on (the ring is kept
        # RING deep: mostly queueing) and (b) one frame at a time (oatgpu_track_batch_dev: what a camera-paced tracker sees)
        "latency_us": dict(
            saturated_p50=_dig(tr, "saturated_latency_us", "p50"), saturated_p99=_dig(tr, "saturated_latency_us", "p99"),
            ring_depth=RING, single_p50=(one_lat or {}).get("p50"), single_p99=(one_lat or {}).get("p99")),
        "early_blob_timeouts": early_timeouts,
        "device_open_retries": (sum(q.get("device_open_retries", 0) for q in per_rank) if per_rank else open_retries),
        "rccl": ({"ranks": world, "backend": args.backend,
                  "version": ".".join(str(x) for x in torch.cuda.nccl.version()) if args.backend == "nccl" else None}
                 if world > 1 else None),
        "scatter_ingest": None,
        "positions_found": n_found,
        "positions_expected": total_streams * tr["steps_timed"],
        # ... of which frames that carry a target: pool frame 0 (it initialises the models) has no disc and recurs once a pool cycle
        "positions_with_target": world * tr["with_target"],
        "parity": parity,
        "parity_detail": parity_detail,
        "input": args.input,
        "dense_model": bool(args.dense_model),
    }

    if solo and not args.no_extra and args.input == "device" and not args.dense_model:
        line["extra_workloads"] = extra
        line["one_frame_a_launch"] = one_frame
        line["default_learning_rate_0"] = frozen

    line["pipeline"] = None
    if solo and not args.no_pipeline and args.input == "device" and not args.dense_model:
        try:
            t0 = time.perf_counter()
            line["pipeline"] = pipeline_block(local_rank)
            log(f"pipeline block: {time.perf_counter() - t0:.1f} s")
        except Exception as e:
            log("pipeline block failed:", e)
    if not args.no_cpu_baseline and solo:          # rank 0 at N = 1 only (the other ranks would idle meanwhile)
        line["cpu_baseline"] = cpu_baseline(args.workload, pool_host0)
    else:
        line["cpu_baseline"] = None
    if want_scatter or extra_fn:                   # last: nothing the line needs from the other ranks is still outstanding
        sc_ = scatter_with_watchdog(args, world, rank, dev, reduce_max, line, t_start, also=other if extra_fn else None,
                                    extra_fn=extra_fn, want_scatter=want_scatter)
        if want_scatter:
            line["scatter_ingest"] = sc_
        sc = line["scatter_ingest"] or {}
        log(f"scatter_ingest: {sc.get('fps')} fps, {sc.get('ms_per_step')} ms/step, {sc.get('bytes_per_peer')} B/peer/step, parity {sc.get('parity')}")
    line["bench_wall_s"] = time.perf_counter() - t_start
    emit(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
