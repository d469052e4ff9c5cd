#!/usr/bin/env python3
"""bench.py -- throughput of the fused hot path (mog + hsv + erode/dilate + blob) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one frame for every camera stream a rank owns, through the whole
fused chain (MOG2 update, setTo, BGR2HSV, inRange, erode, dilate, labelling,
contour sums, selection, result D2H).  Input frames are synthetic, uchar3, and
already resident in HBM when the timed region starts.  Streams are independent,
so N > 1 shards streams over ranks with no data-path collective ("weak"
scaling: per-GPU work fixed); rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs):
    1080p1   1 x 1920x1080 stream per GPU            (configs[1], the default)
    1080p16  16 x 1920x1080 streams batched per GPU  (configs[2]; configs[3] at N=8 is 8/GPU)
    1080p8   8 x 1920x1080 per GPU                   (configs[3] shard)
    4k1      1 x 3840x2160 per GPU, erode 7 dilate 7 (configs[4])
    vga1     1 x 640x480                             (configs[0] shape)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_PIXEL = 205           # 3 B BGR + 101 B model read + 101 B model write

WORKLOADS = {
    "1080p1": dict(rows=1080, cols=1920, streams=1, erode=3, dilate=7),
    "1080p8": dict(rows=1080, cols=1920, streams=8, erode=3, dilate=7),
    "1080p16": dict(rows=1080, cols=1920, streams=16, erode=3, dilate=7),
    "4k1": dict(rows=2160, cols=3840, streams=1, erode=7, dilate=7),
    "vga1": dict(rows=480, cols=640, streams=1, erode=3, dilate=7),
}
ALPHA = 0.01
RESTORE = 1                     # MOG2Invoker's `nmodes = nNewModes;` (oracle/mog2.c "Mode count"); 0 = the other reading
AREA = (20.0, 1e5)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_hotpath(wl, device, ring_depth, dense=False):
    import oat_amd
    from oat_amd.synth import disc_hsv_window
    win = dict(h_thresh=(0, 256), s_thresh=(0, 256), v_thresh=(255, 256)) if dense else disc_hsv_window()
    return oat_amd.HotPath(wl["rows"], wl["cols"], n_streams=wl["streams"], adaptation_coeff=ALPHA,
                           erode=wl["erode"], dilate=wl["dilate"], area=AREA, device=device,
                           ring_depth=ring_depth, mog_restore_nmodes=RESTORE, **win)


def oracle_params(wl):
    import oracle_lib as O
    return O.hsv_params(h_lo=100, h_hi=125, s_lo=150, s_hi=256, v_lo=100, v_hi=256, erode=wl["erode"],
                        dilate=wl["dilate"], min_area=AREA[0], max_area=AREA[1])


def parity_gate(wl, device, frames_seq):
    """masks pixel-exact and centroids identical vs the oracle, on a short fresh run
    of stream 0's frames (the oracle only CHECKS here; it is never the thing measured)."""
    import oracle_lib as O
    import oat_amd
    from oat_amd.synth import disc_hsv_window
    hp = oat_amd.HotPath(wl["rows"], wl["cols"], n_streams=1, adaptation_coeff=ALPHA, erode=wl["erode"],
                         dilate=wl["dilate"], area=AREA, device=device, mog_restore_nmodes=RESTORE, **disc_hsv_window())
    orc = O.Mog2(wl["rows"], wl["cols"], 3, params=dict(restore_nmodes=RESTORE))
    p = oracle_params(wl)
    for t, f in enumerate(frames_seq):
        got = hp.track([f])[0]
        want, thr = O.chain_step(orc, f, ALPHA, p, nthreads=os.cpu_count() or 1)
        if not (hp.read_mask(1) == thr).all():
            return f"mask mismatch at frame {t}"
        if got.position_valid != want["valid"]:
            return f"valid mismatch at frame {t}"
        if want["valid"] and (abs(got.x - want["x"]) > 1e-4 or abs(got.y - want["y"]) > 1e-4):
            return f"centroid mismatch at frame {t}"
    hp.close()
    return "ok"


def measured_run_gate(wl, frames_of_step, got_positions, nthreads):
    """SURVEY.md 8d 'parity gates run with every benchmark': the positions the TIMED run itself
    produced for stream 0, step by step, against the oracle chain run over the very same frame
    sequence (model init, warm-up, timed steps).  frames_of_step: host frames of stream 0 in run
    order; got_positions: (step index in that order, Position2D) pairs to compare."""
    import oracle_lib as O
    orc = O.Mog2(wl["rows"], wl["cols"], 3, params=dict(restore_nmodes=RESTORE))
    p = oracle_params(wl)
    want = [O.chain_step(orc, f, ALPHA, p, nthreads=nthreads)[0] for f in frames_of_step]
    for t, g in got_positions:
        w = want[t]
        if g.position_valid != w["valid"]:
            return f"valid mismatch at run frame {t}"
        if w["valid"] and ((g.a00, g.a10, g.a01) != (w["a00"], w["a10"], w["a01"]) or
                           abs(g.x - w["x"]) > 1e-4 or abs(g.y - w["y"]) > 1e-4):
            return f"centroid mismatch at run frame {t}"
    return "ok"


def cpu_baseline(wl, frames_seq, budget_s=12.0):
    """The oracle (a port of the reference's CPU chain) timed on this host, bounded sample."""
    import oracle_lib as O
    # the port spawns its row workers per stage (no pool): beyond ~32 threads creation cost eats the gain
    ncores = min(os.cpu_count() or 1, 32)
    orc = O.Mog2(wl["rows"], wl["cols"], 3, params=dict(restore_nmodes=RESTORE))
    p = oracle_params(wl)
    O.chain_step(orc, frames_seq[0], ALPHA, p, nthreads=ncores)      # frame 1 (model init), untimed
    n = 0
    t0 = time.perf_counter()
    while True:
        O.chain_step(orc, frames_seq[(n + 1) % len(frames_seq)], ALPHA, p, nthreads=ncores)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 2000:
            break
    # (a) of SURVEY.md 8d: the same chain on ONE thread, a shorter sample
    n1 = 0
    t1 = time.perf_counter()
    while True:
        O.chain_step(orc, frames_seq[(n1 + 1) % len(frames_seq)], ALPHA, p, nthreads=1)
        n1 += 1
        el1 = time.perf_counter() - t1
        if el1 >= budget_s / 4 or n1 >= 500:
            break
    return dict(value=n / el, unit="frames/s", cores=ncores, kind="port",
                sample=f"{n} frames of one {wl['cols']}x{wl['rows']} stream, {el:.1f} s, oracle chain "
                       f"(MOG2, HSV, inRange, morphology rows over {ncores} threads; contour following 1 thread)",
                value_1thread=n1 / el1, sample_1thread=f"{n1} frames, {el1:.1f} s, 1 thread")


def make_pool_dense(rows, cols, ns, nframes, rank, dev):
    """Worst case for the model traffic: every pixel jumps among six well separated colours, so all
    five mixture modes stay live and every plane is read and written every frame (205 B/px real)."""
    g = torch.Generator(device=dev)
    g.manual_seed(0xD0 + rank)
    table = torch.tensor([[20, 30, 40], [90, 200, 60], [200, 60, 120], [240, 240, 230], [40, 130, 220], [140, 20, 150]],
                         device=dev, dtype=torch.int16)
    pool = []
    for t in range(nframes):
        idx = torch.randint(0, 6, (ns, rows, cols), device=dev, generator=g)
        f = table[idx] + torch.randint(-5, 6, (ns, rows, cols, 3), device=dev, dtype=torch.int16, generator=g)
        pool.append(f.clamp_(0, 255).to(torch.uint8).contiguous())
    return pool


def make_pool_device(rows, cols, ns, nframes, rank, dev):
    """[nframes] tensors of shape (ns, rows, cols, 3) uint8 on `dev`."""
    from oat_amd.synth import DISC_BGR
    g = torch.Generator(device=dev)
    g.manual_seed(0x0A7 + rank)
    yy = torch.arange(rows, device=dev, dtype=torch.float32).view(rows, 1)
    xx = torch.arange(cols, device=dev, dtype=torch.float32).view(1, cols)
    base = 110 + 30 * (xx / max(cols - 1, 1)) + 20 * (yy / max(rows - 1, 1))
    base = torch.stack([base, base + 6, base - 5], -1).to(torch.int16)            # (rows, cols, 3)
    flick = ((torch.arange(rows * cols, device=dev) % 64) == 17).view(rows, cols)
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
    pool = []
    for t in range(nframes):
        f = base.unsqueeze(0) + torch.randint(-6, 7, (ns, rows, cols, 3), device=dev, dtype=torch.int16, generator=g)
        f[:, flick] += 60 if (t & 1) else -50
        f = f.clamp_(0, 255).to(torch.uint8)
        if t > 0:                                   # frame 0 (model initialisation) has no discs
            for s in range(ns):
                for d in discs[s]:
                    cx = int(round(cols / 2 + d["ax"] * np.sin(d["fx"] * 9 * t + d["px"])))
                    cy = int(round(rows / 2 + d["ay"] * np.sin(d["fy"] * 9 * t + d["py"])))
                    r = d["r"]
                    y0, y1, x0, x1 = max(cy - r, 0), min(cy + r + 1, rows), max(cx - r, 0), min(cx + r + 1, cols)
                    m = ((xx[:, x0:x1] - cx) ** 2 + (yy[y0:y1] - cy) ** 2) <= r * r
                    sub = f[s, y0:y1, x0:x1]
                    sub[m] = torch.tensor(d["col"], device=dev, dtype=torch.uint8)
        pool.append(f.contiguous())
    return pool


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="1080p1", choices=sorted(WORKLOADS))
    ap.add_argument("--pool", type=int, default=48, help="distinct frame sets resident in HBM")
    ap.add_argument("--input", default="device", choices=["device", "host", "host-sync"],
                    help="device: frames resident in HBM (the headline); host: pageable host frames through "
                         "oatgpu_track_batch, i.e. PCIe-inclusive (reported in DESIGN.md, never the headline)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to smoke-test the "
                         "multi-rank control flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--dense-model", action="store_true",
                    help="diagnostic: input that keeps all 5 mixture modes live on every pixel (K1 moves the full "
                         "205 B/px) and a threshold window nothing passes; not a BASELINE config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-step-calls", action="store_true",
                    help="drive the pipelined path with one enqueue and one collect call per step from Python "
                         "instead of one oatgpu_track_sequence_dev call for the whole run")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--check-steps", type=int, default=64,
                    help="timed steps of stream 0 replayed through the oracle after the run (0 = off)")
    ap.add_argument("--learning-rate", type=float, default=None,
                    help="MOG2 adaptation coefficient (default 0.01 = SURVEY 8d; 0 = Oat's default, frozen model)")
    ap.add_argument("--mog-restore-nmodes", type=int, default=1, choices=[0, 1],
                    help="1 (default): MOG2Invoker's `nmodes = nNewModes;` -- pruned modes keep their slot; 0: pruning "
                         "shrinks the mode count (round 1's reading); oracle/mog2.c 'Mode count'")
    args = ap.parse_args()
    global ALPHA, RESTORE
    RESTORE = args.mog_restore_nmodes
    if args.learning_rate is not None:
        ALPHA = args.learning_rate

    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if (world == 1 or args.backend == "nccl") else torch.device("cpu")

    wl = WORKLOADS[args.workload]
    rows, cols, ns = wl["rows"], wl["cols"], wl["streams"]
    K, W = args.steps, args.warmup
    ring = 8

    # ---- synthetic input pool, generated on the device once (this rank's streams have global
    # ids rank*ns ..): gradient + fresh +-6 noise per frame, every 64th pixel flickering, two
    # saturated discs per stream on Lissajous paths.  The pool is long enough that
    # a disc revisits a pixel too rarely to be learnt as background.
    pool = (make_pool_dense if args.dense_model else make_pool_device)(rows, cols, ns, args.pool, rank, dev)
    torch.cuda.synchronize()
    pool_host = [p[0:1].cpu().numpy() for p in pool[:8]]     # stream 0, for the parity gate / CPU baseline

    hp = make_hotpath(wl, local_rank, ring, dense=args.dense_model)

    def barrier():
        hp.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    positions = []
    raw_results = []

    host_pool = None
    if args.input in ("host", "host-sync"):
        # PCIe-inclusive variants (never the headline value): "host" = page-locked frames through the
        # pipelined oatgpu_track_enqueue (copies overlap compute), "host-sync" = pageable frames
        # through the synchronous oatgpu_track_batch (what a one-frame-at-a-time caller gets)
        if args.input == "host":
            host_pool = [[f.numpy() for f in p.cpu().pin_memory()] for p in pool]
        else:
            host_pool = [[np.ascontiguousarray(f) for f in p.cpu().numpy()] for p in pool]

    prepared, sequence_out = {}, []

    def prepare(nsteps):
        import ctypes as C
        from oat_amd import ffi
        if host_pool is None and not args.per_step_calls:
            prepared[nsteps] = ((C.c_void_p * nsteps)(*[pool[(i + 1) % len(pool)].data_ptr() for i in range(nsteps)]),
                                (ffi.Position * (nsteps * ns))())

    def run(nsteps, keep=False):
        if host_pool is not None and args.input == "host-sync":
            for i in range(nsteps):
                r = hp.track(host_pool[(i + 1) % len(host_pool)])
                if keep:
                    positions.append(r)
            return
        import ctypes as C
        from oat_amd import ffi
        lib, ctx, lr = hp.lib, hp.ctx, hp.learning_coeff_
        if host_pool is None and not args.per_step_calls:
            # Device-resident frames: the whole sequence through oatgpu_track_sequence_dev, i.e. the
            # enqueue/collect loop inside the library -- at ~35 us per step two Python->C calls per step
            # are a measurable part of it (--per-step-calls times them from Python instead).  The
            # argument arrays are built by prepare() outside the timed region.
            seq, out = prepared.pop(nsteps)
            ffi.check(lib, ctx, lib.oatgpu_track_sequence_dev(ctx, seq, nsteps, lr, out))
            if keep:
                sequence_out.append((out, nsteps))
            return
        # Thin loop straight on the C ABI; the raw result records are kept and converted after the
        # timed region (a dataclass per position would be a measurable part of a step).
        enq, col = lib.oatgpu_track_enqueue_dev, lib.oatgpu_track_collect
        if host_pool is not None:
            enq_h = lib.oatgpu_track_enqueue
            ptrs = [(ffi._u8p * ns)(*[ffi.u8(f) for f in fs]) for fs in host_pool]
            enq = lambda ctx_, p_, lr_: enq_h(ctx_, p_, ns, lr_)
        else:
            ptrs = [C.c_void_p(p.data_ptr()) for p in pool]
        npool = len(ptrs)
        bufs = [(ffi.Position * ns)() for _ in range(nsteps)] if keep else [(ffi.Position * ns)()]
        outstanding = got = 0
        for i in range(nsteps):
            if outstanding == ring:
                ffi.check(lib, ctx, col(ctx, bufs[got if keep else 0]))
                got += 1
                outstanding -= 1
            ffi.check(lib, ctx, enq(ctx, ptrs[(i + 1) % npool], lr))
            outstanding += 1
        while outstanding:
            ffi.check(lib, ctx, col(ctx, bufs[got if keep else 0]))
            got += 1
            outstanding -= 1
        if keep:
            raw_results.extend(bufs)

    # frame 1 initialises the models with the disc-free frame, then warm-up
    hp.track_dev(pool[0].data_ptr())
    prepare(W)
    run(W)
    hp.profile(16)           # HIP events around K1 on every 16th step of the timed region
    hp.profile_reset()
    prepare(K)
    barrier()
    t0 = time.perf_counter()
    run(K, keep=True)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = hp.profile_read()
    hp.profile(0)
    from oat_amd.components import Position2D
    positions.extend([Position2D.from_c(p) for p in buf] for buf in raw_results)
    for out, nsteps in sequence_out:
        positions.extend([Position2D.from_c(out[t * ns + s_]) for s_ in range(ns)] for t in range(nsteps))

    # parity gate on this run's own frames -- after the timed region, with fresh
    # contexts, so that it cannot disturb the measurement
    parity = "skipped"
    parity_detail = None
    if rank == 0 and not args.no_parity and not args.dense_model:
        parity = parity_gate(wl, local_rank, [p[0] for p in pool_host[:4]])
        log("parity gate:", parity)
        if parity == "ok" and args.check_steps > 0:
            # the timed run's own output: model init frame, W warm-up frames, then the first
            # check_steps timed frames of stream 0, replayed through the oracle
            G = min(args.check_steps, K)
            s0 = [p[0].cpu().numpy() for p in pool]                  # stream 0 of every pool frame
            order = [0] + [(i + 1) % len(pool) for i in range(W)] + [(i + 1) % len(pool) for i in range(G)]
            parity = measured_run_gate(wl, [s0[i] for i in order], [(1 + W + i, positions[i][0]) for i in range(G)],
                                       min(os.cpu_count() or 1, 32))
            log(f"measured-run gate ({G} timed steps of stream 0 vs oracle):", parity)
            parity_detail = f"fresh-context masks+centroids on 4 frames; first {G} timed steps of stream 0 vs the oracle: {parity}"

    # achievable HBM rates of this very device (plain streaming kernels), rank 0 only, after the timed region
    hbm_read = hbm_copy = None
    if rank == 0:
        try:
            hbm_read, hbm_copy = hp.measure_hbm(1 << 30, 5)
        except Exception as e:          # never let the probe break the benchmark line
            log("hbm probe failed:", e)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        found = torch.tensor([sum(p.position_valid for r in positions for p in r)], dtype=torch.int64, device=red_dev)
        dist.all_reduce(found, op=dist.ReduceOp.SUM)
        n_found = int(found.item())
    else:
        n_found = sum(p.position_valid for r in positions for p in r)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # real HBM bytes per K1 launch from the committed PMC passes (profiles/collect_pmc.sh writes
    # the file; the counters cannot be read from inside this process)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            tj = json.load(f)
        if args.dense_model:        # the calibration pass of collect_pmc.sh IS the dense 4K run
            cal = tj.get("_calibration", {})
            if args.workload == "4k1" and cal:
                traffic = cal["fetch_factor"] * cal["FETCH_SIZE_KiB"] * 1024 + cal["WRITE_SIZE_KiB"] * 1024
        else:
            traffic = tj.get(args.workload, {}).get("k_mog_fused_bytes_per_launch")
    except Exception:
        traffic = None

    total_streams = ns * world
    fps = total_streams * K / elapsed
    px_per_launch = rows * cols * ns
    # HIP-event time of the K1 launch on its own stream, minus what an EMPTY event pair measures
    # there (calibrated by the library): that is the kernel's duration as rocprofv3 reports it.
    mog_ms_raw = prof["mog_ms"] / max(prof["steps"], 1)
    mog_ms = max(mog_ms_raw - prof["event_pair_ms"], 1e-6)
    achieved = BYTES_PER_PIXEL * px_per_launch / (mog_ms * 1e-3) / 1e9 if mog_ms > 0 else 0.0
    line = {
        "metric": "frames/sec/GPU (1080p & 4K) mog+hsv+ccl fused; % HBM roofline",
        "value": fps,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{ns} x {cols}x{rows} uchar3 stream(s) per GPU, MOG2(5 mixtures, lr {ALPHA}) + HSV + "
                               f"inRange + erode {wl['erode']} + dilate {wl['dilate']} + external-contour centroid",
                   "name": args.workload, "streams_per_gpu": ns, "rows": rows, "cols": cols,
                   "learning_rate": ALPHA, "parallelism": f"streams sharded, {world} rank(s)"},
        "fps_per_gpu": fps / world,
        "hbm_roofline_frac_whole_step": BYTES_PER_PIXEL * px_per_launch / (elapsed / K) / 1e9 / HBM_PEAK_GBPS,
        "roofline": {"bound": "hbm", "kernel": "k_mog_fused", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "note": "achieved = ALGORITHMIC 205 B/px / K1 time; K1 skips planes of dead modes and "
                             "unchanged planes, so real traffic (PMC) is below algorithmic and frac may exceed 1",
                     "bytes_per_launch": BYTES_PER_PIXEL * px_per_launch, "avg_launch_ms": mog_ms,
                     "measured_stream_read_GBps": hbm_read, "measured_stream_copy_GBps": hbm_copy,
                     "traffic_GBps": (traffic / (mog_ms * 1e-3) / 1e9) if traffic else None,
                     "avg_launch_ms_raw_events": mog_ms_raw, "empty_event_pair_ms": prof["event_pair_ms"]},
        "stage_ms": {"mog": mog_ms, "morph": prof["morph_ms"] / max(prof["steps"], 1),
                     "blob": prof["blob_ms"] / max(prof["steps"], 1),
                     "gpu_total": prof["total_ms"] / max(prof["steps"], 1)},
        "positions_found": n_found,
        "positions_expected": total_streams * K,
        "parity": parity,
        "parity_detail": parity_detail,
        "input": args.input,
        "dense_model": bool(args.dense_model),
    }
    if not args.no_cpu_baseline and world == 1:          # rank 0 at N = 1 only (the other ranks would idle meanwhile)
        line["cpu_baseline"] = cpu_baseline(wl, [p[0] for p in pool_host])
    else:
        line["cpu_baseline"] = None
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
