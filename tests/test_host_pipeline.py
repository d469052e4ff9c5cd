"""The C++ drop-in side: shm transport protocol (CPU) and the component pipeline (GPU).

BASELINE config 1 plumbing:  frameserve -> framefilt mog -> framefilt col -C HSV -> posidet hsv,
one OS process per component over named POSIX shm nodes, plus the fused single-process
oat-track-hip; positions must equal the CPU oracle chain frame by frame, and every token must
carry its frame's sample count (PositionDetector.cpp:80)."""
import json
import os
import subprocess
import time
import uuid

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "build", "bin")


@pytest.fixture(scope="module")
def host_bins():
    subprocess.check_call(["make", "-s", "-j4", "-C", ROOT, "host"])
    return BIN


def test_shm_protocol_scenarios_and_wire_formats(host_bins, tmp_path):
    """Reference test/shmemdf/*_test.cpp scenarios restated in oat_amd/host/test_shmemdf.cpp, plus the
    position wire formats: the JSON serialiser (Position2D.h:170-233) and the packed 82-byte record
    (Position2D.cpp:24-96), which numpy must read back with the reference's NPY dtype."""
    npy = tmp_path / "pos.npy"
    out = subprocess.run([os.path.join(host_bins, "test_shmemdf"), "oat_t_" + uuid.uuid4().hex[:8], str(npy)],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failures" in out.stdout
    a = np.load(npy)
    assert a.dtype.itemsize == 82 and a.dtype.names == ("tick", "usec", "unit", "pos_ok", "pos_xy", "vel_ok",
                                                         "vel_xy", "head_ok", "head_xy", "reg_ok", "reg")
    assert a["tick"].tolist() == [3, 4, 5] and a["usec"].tolist() == [30000, 40000, 50000]
    assert a["pos_ok"].tolist() == [1, 0, 1] and a["pos_xy"].tolist() == [[0.0, 0.0], [10.5, -1.0], [21.0, -2.0]]
    assert a["vel_ok"].tolist() == [0, 0, 0] and a["unit"].tolist() == [0, 0, 0]


def test_binaries_exist_and_print_usage(host_bins):
    for b in ("oat-framefilt-hip", "oat-posidet-hip", "oat-track-hip", "oat-frameserve-raw", "oat-posi-cout"):
        r = subprocess.run([os.path.join(host_bins, b), "--help"], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and "Usage" in r.stdout, (b, r.stdout, r.stderr)
    # wrong TYPE -> the reference's message and exit code -1 (framefilter/main.cpp:200, :278-295)
    r = subprocess.run([os.path.join(host_bins, "oat-posidet-hip"), "nope", "a", "b"], capture_output=True, text=True)
    assert r.returncode == 255 and "Selected TYPE is invalid." in r.stderr


def test_ingest_root_arguments_are_checked_before_any_device_is_touched(host_bins):
    """`oat-track-hip --ingest-root` (scatter_tracker.hpp): the root must be one of --gpu-index, every device once (one RCCL
    rank per device), and the options of the other fused stages are refused -- exit code -1 and the reason on stderr
    (framefilter/main.cpp:278-295), no GPU needed to say so."""
    B = os.path.join(host_bins, "oat-track-hip")
    for extra, msg in ((["--gpu-index", "0,1", "--ingest-root", "2"], "root device must be one of --gpu-index"),
                       (["--gpu-index", "0,0", "--ingest-root", "0"], "every device of --gpu-index once"),
                       (["--gpu-index", "0", "--ingest-root", "0", "--kalman"], "--ingest-root does not take --kalman")):
        r = subprocess.run([B, "a,b", "c,d"] + extra, capture_output=True, text=True, timeout=60)
        assert r.returncode == 255 and msg in r.stderr, (extra, r.returncode, r.stderr)
    r = subprocess.run(["ldd", B], capture_output=True, text=True)
    assert "librccl" in r.stdout and "libamdhip64" in r.stdout and "liboatgpu" in r.stdout, r.stdout      # RCCL behind the C++ boundary


@pytest.mark.parametrize("form", ["per-rank ingest", "ingest-root"])
def test_the_cxx_launchers_partition_is_dist_pys(host_bins, form):
    """camera s lives on shard s // ceil(S / N) for life.  `oat-track-hip --print-partition` prints what BOTH C++ launch
    forms do with a SOURCE list and a device list -- no device touched -- and it must be oat_amd.dist.stream_partition (what
    bench.py's ranks own) for every S and N up to BASELINE configs[3]'s 64 cameras on 8 devices, more devices than cameras
    included."""
    from oat_amd.dist import stream_partition
    B = os.path.join(host_bins, "oat-track-hip")
    for S, N in [(1, 1), (3, 2), (8, 8), (9, 8), (16, 8), (64, 8), (5, 8), (7, 3), (2, 4)]:
        devs = ",".join(str(d) for d in range(N))
        cmd = [B, ",".join(f"c{i}" for i in range(S)), ",".join(f"p{i}" for i in range(S)), "--gpu-index", devs, "--print-partition"]
        if form == "ingest-root":
            cmd += ["--ingest-root", "0"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, (S, N, r.stderr)
        got = [l.split() for l in r.stdout.splitlines() if l.startswith("shard")]
        want = [(k, stream_partition(S, N, k)) for k in range(N) if len(stream_partition(S, N, k))]
        assert [(int(g[1]), int(g[3]), int(g[5]), int(g[6])) for g in got] == [(k, k, b.start, b.stop) for k, b in want], (S, N, r.stdout)
        if form == "ingest-root":
            assert [g[7] for g in got] == ["root"] + ["peer"] * (len(got) - 1), r.stdout


def test_feeder_to_reader_over_shm_without_gpu(host_bins, tmp_path):
    """Token discipline of the transport alone: N frames in -> N tokens seen, in order."""
    # a Position2D-typed reader must refuse a Frame node (Source<T> type check)
    rows, cols, n = 4, 6, 5
    raw = tmp_path / "f.raw"
    np.arange(rows * cols * 3, dtype=np.uint8).tofile(raw)
    addr = "oat_t_" + uuid.uuid4().hex[:8]
    reader = subprocess.Popen([os.path.join(host_bins, "oat-posi-cout"), addr], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True)
    time.sleep(0.3)
    feeder = subprocess.Popen([os.path.join(host_bins, "oat-frameserve-raw"), addr, "-f", str(raw), "--rows",
                               str(rows), "--cols", str(cols), "-n", str(n)], stderr=subprocess.PIPE, text=True)
    out, err = reader.communicate(timeout=30)
    feeder.wait(timeout=30)
    assert reader.returncode == 255 and "Type mismatch" in err
    subprocess.run([os.path.join(host_bins, "oat-clean-hip"), addr], capture_output=True)


def _consumers_ready(*addresses, timeout=30.0, settle=1.0):
    """Every consumer touch()es the node of its SOURCE address when it starts (Source.h:118-139), and a token served before
    that is a token the consumer never sees.  A fixed sleep was the test's assumption about process start-up time (one
    failure on a slow box, r05q): wait for the node segments "/dev/shm/<addr>_node" instead, then a moment for the touch
    itself."""
    t_end = time.monotonic() + timeout
    while time.monotonic() < t_end and not all(os.path.exists(f"/dev/shm/{a}_node") for a in addresses):
        time.sleep(0.05)
    time.sleep(settle)


def _run_pipeline(host_bins, tmp_path, frames, fused, mog_args=(), config=None):
    rows, cols = frames[0].shape[:2]
    raw = tmp_path / "frames.raw"
    np.stack(frames).tofile(raw)
    tag = "oat_t_" + uuid.uuid4().hex[:8]
    a_raw, a_filt, a_hsv, a_pos = (tag + s for s in ("raw", "filt", "hsv", "pos"))
    det = ["-H", "[100,125]", "-S", "[150,256]", "-V", "[100,256]", "-e", "3", "-d", "7"]
    if config:                       # the same options from table [KEY] of a TOML file (-c FILE KEY)
        det = ["-c", str(config[0]), config[1]]
    procs = []
    B = lambda n: os.path.join(host_bins, n)
    reader = subprocess.Popen([B("oat-posi-cout"), a_pos], stdout=subprocess.PIPE, text=True)
    if fused:
        procs.append(subprocess.Popen([B("oat-track-hip"), a_raw, a_pos, "-a", "0.01", "--area", "[20,100000]"] + det + list(mog_args)))
    else:
        procs.append(subprocess.Popen([B("oat-posidet-hip"), "hsv", a_hsv, a_pos, "-a", "[20,100000]"] + det))
        procs.append(subprocess.Popen([B("oat-framefilt-hip"), "col", a_filt, a_hsv, "-C", "HSV"]))
        procs.append(subprocess.Popen([B("oat-framefilt-hip"), "mog", a_raw, a_filt, "-a", "0.01"] + list(mog_args)))
    _consumers_ready(a_pos, a_raw, *(() if fused else (a_filt, a_hsv)))   # every consumer has touch()ed its node before the first token exists
    feeder = subprocess.Popen([B("oat-frameserve-raw"), a_raw, "-f", str(raw), "--rows", str(rows), "--cols",
                               str(cols), "-n", str(len(frames)), "-r", "200"])
    try:
        out, _ = reader.communicate(timeout=180)
        feeder.wait(timeout=60)
        for p in procs:
            p.wait(timeout=60)
    finally:
        for p in procs + [feeder, reader]:
            if p.poll() is None:
                p.kill()
        subprocess.run([B("oat-clean-hip"), a_raw, a_filt, a_hsv, a_pos], capture_output=True)
    assert all(p.returncode == 0 for p in procs), [p.returncode for p in procs]
    return [json.loads(l) for l in out.splitlines() if l.strip()]


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_component_pipeline_matches_oracle(host_bins, tmp_path, fused):
    import oracle_lib as O
    from oat_amd.synth import SyntheticStream
    rows, cols, n = 480, 640, 24
    st = SyntheticStream(rows, cols, 3, n_discs=2)
    frames = [st.frame(t, with_discs=t > 0) for t in range(n)]
    got = _run_pipeline(host_bins, tmp_path, frames, fused)
    assert len(got) == n                                   # exactly one token out per token in
    orc = O.Mog2(rows, cols, 3)
    p = O.hsv_params(h_lo=100, h_hi=125, s_lo=150, s_hi=256, v_lo=100, v_hi=256, erode=3, dilate=7,
                     min_area=20.0, max_area=1e5)
    hits = 0
    for t, (f, g) in enumerate(zip(frames, got)):
        want, _ = O.chain_step(orc, f, 0.01, p)
        assert g["tick"] == t + 1 and g["usec"] == (t + 1) * 5000      # Sample propagated (200 Hz)
        assert g["pos_ok"] == want["valid"], t
        if want["valid"]:
            hits += 1
            assert abs(g["pos_xy"][0] - want["x"]) < 1e-4 and abs(g["pos_xy"][1] - want["y"]) < 1e-4, t
    assert hits >= n - 2


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_model_file_resumes_across_process_restarts(host_bins, tmp_path, fused):
    """--model-file: the pipeline is stopped after 12 frames and started again; the restarted
    processes continue from the checkpointed MOG2 model exactly where the oracle (never stopped) is."""
    import oracle_lib as O
    from oat_amd.synth import SyntheticStream
    rows, cols, n = 240, 320, 24
    st = SyntheticStream(rows, cols, 5, n_discs=1)
    frames = [st.frame(t, with_discs=t > 0) for t in range(n)]
    model = tmp_path / "bg.mog"
    args = ["--model-file", str(model)]
    got = _run_pipeline(host_bins, tmp_path, frames[:12], fused, args)
    assert model.exists() and model.stat().st_size == 64 + rows * cols * 101
    got += _run_pipeline(host_bins, tmp_path, frames[12:], fused, args)
    assert len(got) == n
    orc = O.Mog2(rows, cols, 3)
    p = O.hsv_params(h_lo=100, h_hi=125, s_lo=150, s_hi=256, v_lo=100, v_hi=256, erode=3, dilate=7,
                     min_area=20.0, max_area=1e5)
    hits = 0
    for t, (f, g) in enumerate(zip(frames, got)):
        want, _ = O.chain_step(orc, f, 0.01, p)
        assert g["pos_ok"] == want["valid"], t
        if want["valid"]:
            hits += 1
            assert abs(g["pos_xy"][0] - want["x"]) < 1e-4 and abs(g["pos_xy"][1] - want["y"]) < 1e-4, t
    assert hits >= n - 3


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_grey_component_pipeline_matches_oracle(host_bins, tmp_path, fused):
    """The thresh chain from a BGR camera, one OS process per component as the reference runs it:
    frameserve -> framefilt col -C GREY -> framefilt mog -> posidet thresh (SimpleThreshold.cpp:46 wants GREY
    frames, Source.h:300-313 refuses anything else "Maybe use oat-framefilt col?")."""
    import oracle_lib as O
    from oat_amd.synth import SyntheticStream
    rows, cols, n = 240, 320, 20
    st = SyntheticStream(rows, cols, 4, n_discs=1, radius=14)
    frames = [st.frame(t, with_discs=t > 0) for t in range(n)]
    raw = tmp_path / "frames.raw"
    np.stack(frames).tofile(raw)
    tag = "oat_t_" + uuid.uuid4().hex[:8]
    a_raw, a_grey, a_filt, a_pos = (tag + s for s in ("raw", "grey", "filt", "pos"))
    B = lambda x: os.path.join(host_bins, x)
    reader = subprocess.Popen([B("oat-posi-cout"), a_pos], stdout=subprocess.PIPE, text=True)
    if fused:
        procs = [subprocess.Popen([B("oat-track-hip"), a_grey, a_pos, "-a", "0.01", "--thresh", "[30,100]", "-e", "3",
                                   "-d", "7", "--area", "[20,100000]"])]
    else:
        procs = [subprocess.Popen([B("oat-posidet-hip"), "thresh", a_filt, a_pos, "-T", "[30,100]", "-e", "3", "-d", "7",
                                   "-a", "[20,100000]"]),
                 subprocess.Popen([B("oat-framefilt-hip"), "mog", a_grey, a_filt, "-a", "0.01"])]
    procs.append(subprocess.Popen([B("oat-framefilt-hip"), "col", a_raw, a_grey, "-C", "GREY"]))
    _consumers_ready(a_pos, a_raw, a_grey, *(() if fused else (a_filt,)))
    feeder = subprocess.Popen([B("oat-frameserve-raw"), a_raw, "-f", str(raw), "--rows", str(rows), "--cols", str(cols),
                               "-n", str(n), "-r", "200"])
    try:
        out, _ = reader.communicate(timeout=180)
        feeder.wait(timeout=60)
        for p in procs:
            p.wait(timeout=60)
    finally:
        for p in procs + [feeder, reader]:
            if p.poll() is None:
                p.kill()
        subprocess.run([B("oat-clean-hip"), a_raw, a_grey, a_filt, a_pos], capture_output=True)
    assert all(p.returncode == 0 for p in procs), [p.returncode for p in procs]
    got = [json.loads(l) for l in out.splitlines() if l.strip()]
    assert len(got) == n
    orc = O.Mog2(rows, cols, 1)
    prm = O.hsv_params(h_lo=30, h_hi=100, erode=3, dilate=7, min_area=20.0, max_area=1e5)
    hits = 0
    for t, (f, g) in enumerate(zip(frames, got)):
        want, _ = O.chain_step(orc, O.bgr2grey(f), 0.01, prm)
        assert g["tick"] == t + 1 and g["pos_ok"] == want["valid"], t
        if want["valid"]:
            hits += 1
            assert abs(g["pos_xy"][0] - want["x"]) < 1e-4 and abs(g["pos_xy"][1] - want["y"]) < 1e-4, t
    assert hits >= n - 3


def test_col_refuses_what_the_reference_refuses(host_bins):
    """--color is required (ColorConvert.cpp:59-60 -> TOMLSanitize.h:199-200) and must name a colour (Color.h:65-77);
    neither needs a GPU."""
    exe = os.path.join(host_bins, "oat-framefilt-hip")
    r = subprocess.run([exe, "col", "x_src", "x_snk"], capture_output=True, text=True, timeout=30)
    assert r.returncode == 255 and "Required configuration value 'color' was not specified." in r.stderr
    r = subprocess.run([exe, "col", "x_src", "x_snk", "-C", "RGB"], capture_output=True, text=True, timeout=30)
    assert r.returncode == 255 and "Invalid color." in r.stderr


@pytest.mark.gpu
def test_fused_tracker_with_homography_matches_oracle(host_bins, tmp_path):
    """oat-track-hip --homography = ... -> posidet hsv -> posifilt homography in one process: positions in WORLD units
    (JSON "unit": 1, Position2D.h:170-233) equal cv::perspectiveTransform of the oracle chain's pixels."""
    import oracle_lib as O
    from oat_amd.synth import SyntheticStream
    rows, cols, n = 240, 320, 16
    H = [0.01, 0.0, -1.6, 0.0, -0.01, 1.2, 0.0, 1e-4, 1.0]
    st = SyntheticStream(rows, cols, 8, n_discs=1, radius=12)
    frames = [st.frame(t, with_discs=t > 0) for t in range(n)]
    got = _run_pipeline(host_bins, tmp_path, frames, True, ["--homography", "[" + ",".join(repr(v) for v in H) + "]"])
    assert len(got) == n
    orc = O.Mog2(rows, cols, 3)
    p = O.hsv_params(h_lo=100, h_hi=125, s_lo=150, s_hi=256, v_lo=100, v_hi=256, erode=3, dilate=7, min_area=20.0, max_area=1e5)
    hits = 0
    for t, (f, g) in enumerate(zip(frames, got)):
        want, _ = O.chain_step(orc, f, 0.01, p)
        assert g["unit"] == 1 and g["pos_ok"] == want["valid"], t
        if want["valid"]:
            x, y, _, _ = O.homography(H, True, want["x"], want["y"])
            assert abs(g["pos_xy"][0] - x) < 1e-5 and abs(g["pos_xy"][1] - y) < 1e-5, t     # JSON carries 5 decimals
            hits += 1
    assert hits >= n - 3


def test_config_file_errors_are_the_references(host_bins, tmp_path):
    """-c FILE KEY (TOMLSanitize.h:73-118): bad pair, missing table, unknown key -> error exit, the
    reference's messages; none of this needs a GPU (options are parsed before any device work)."""
    toml = tmp_path / "c.toml"
    toml.write_text("[good]\nerode = 3 # comment\nh-thresh = [100,\n 125]\n[bad]\nnokey = 1\n")
    exe = os.path.join(host_bins, "oat-posidet-hip")
    for args, msg in ((["-c", str(toml)], "Configuration must be supplied as file key pair."),
                      (["-c", str(toml), "nope"], "No configuration table named 'nope'"),
                      (["-c", str(toml), "bad"], "Unknown configuration key 'nokey'."),
                      (["-c", str(tmp_path / "absent.toml"), "good"], "Could not open configuration file")):
        r = subprocess.run([exe, "hsv", "x_src", "x_snk"] + args, capture_output=True, text=True, timeout=30)
        assert r.returncode == 255 and msg in r.stderr, (args, r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("via_config", [False, True])
def test_fused_tracker_with_kalman_matches_oracle(host_bins, tmp_path, via_config):
    """oat-track-hip --kalman = frameserve -> mog -> col -> posidet hsv -> posifilt kalman in one process."""
    import oracle_lib as O
    from oat_amd.synth import SyntheticStream
    rows, cols, n = 240, 320, 30
    st = SyntheticStream(rows, cols, 9, n_discs=1)
    frames = [st.frame(t, with_discs=(t > 0 and not 12 <= t < 15)) for t in range(n)]
    if via_config:
        toml = tmp_path / "track.toml"
        toml.write_text("""# everything oat-track-hip needs, from a configuration file
[unrelated]
erode = 11

[tracker]
h-thresh = [100, 125]     # blue disc
s-thresh = [150, 256]
v-thresh = [100,
            256]
erode = 3
dilate = 7
kalman = true
dt = 0.005
timeout = 0.05
sigma-accel = 40.0
sigma-noise = 1
""")
        got = _run_pipeline(host_bins, tmp_path, frames, True, config=(toml, "tracker"))
    else:
        got = _run_pipeline(host_bins, tmp_path, frames, True,
                            ["--kalman", "--dt", "0.005", "-T", "0.05", "--sigma-accel", "40", "-n", "1.0"])
    assert len(got) == n
    orc = O.Mog2(rows, cols, 3)
    kal = O.Kalman(dt=0.005, timeout=0.05, sigma_accel=40.0, sigma_noise=1.0)
    p = O.hsv_params(h_lo=100, h_hi=125, s_lo=150, s_hi=256, v_lo=100, v_hi=256, erode=3, dilate=7,
                     min_area=20.0, max_area=1e5)
    tracked = 0
    for t, (f, g) in enumerate(zip(frames, got)):
        d, _ = O.chain_step(orc, f, 0.01, p)
        k = kal.filter(d["valid"], d["x"], d["y"])
        assert g["pos_ok"] == k["position_valid"] and g["vel_ok"] == k["velocity_valid"], t
        if k["position_valid"]:
            tracked += 1
            assert abs(g["pos_xy"][0] - k["x"]) < 1e-4 and abs(g["pos_xy"][1] - k["y"]) < 1e-4, t
            assert abs(g["vel_xy"][0] - k["vx"]) < 1e-4 * max(1, abs(k["vx"])), t
            assert abs(g["vel_xy"][1] - k["vy"]) < 1e-4 * max(1, abs(k["vy"])), t
    assert tracked >= n - 4


# ------------------------------------------------- batched, pipelined tracker --

def _run_batched(host_bins, tmp_path, streams_frames, extra=(), ring=2, fps=200, tracker_stderr=None):
    """n feeders -> ONE oat-track-hip (n SOURCEs, n SINKs, one context, one launch per stage) -> n readers."""
    n = len(streams_frames)
    rows, cols = streams_frames[0][0].shape[:2]
    tag = "oat_t_" + uuid.uuid4().hex[:8]
    srcs = [f"{tag}raw{s}" for s in range(n)]
    snks = [f"{tag}pos{s}" for s in range(n)]
    B = lambda b: os.path.join(host_bins, b)
    # (readers write to files: pipes drained one after the other would fill up and stall the pipeline)
    files = [open(tmp_path / f"reader{s}.out", "w+") for s in range(n)]
    readers = [subprocess.Popen([B("oat-posi-cout"), a], stdout=f, text=True) for a, f in zip(snks, files)]
    tracker = subprocess.Popen([B("oat-track-hip"), ",".join(srcs), ",".join(snks), "-a", "0.01", "--area", "[20,100000]",
                                "-H", "[100,125]", "-S", "[150,256]", "-V", "[100,256]", "-e", "3", "-d", "7",
                                "--ring", str(ring)] + list(extra), stderr=tracker_stderr)
    _consumers_ready(*srcs, *snks)
    feeders = []
    for s in range(n):
        raw = tmp_path / f"frames{s}.raw"
        np.stack(streams_frames[s]).tofile(raw)
        feeders.append(subprocess.Popen([B("oat-frameserve-raw"), srcs[s], "-f", str(raw), "--rows", str(rows), "--cols",
                                         str(cols), "-n", str(len(streams_frames[s])), "-r", str(fps)]))
    outs = []
    try:
        for r, fl in zip(readers, files):
            r.wait(timeout=240)
            fl.seek(0)
            outs.append([json.loads(l) for l in fl.read().splitlines() if l.strip()])
        for f in feeders:
            f.wait(timeout=60)
        tracker.wait(timeout=60)
    finally:
        for p in readers + feeders + [tracker]:
            if p.poll() is None:
                p.kill()
        subprocess.run([B("oat-clean-hip")] + srcs + snks, capture_output=True)
    assert tracker.returncode == 0
    return outs


@pytest.mark.gpu
@pytest.mark.parametrize("ring", [2, 4])
def test_batched_tracker_four_cameras_match_their_oracles(host_bins, tmp_path, ring):
    """BASELINE configs 3/4 from the drop-in boundary.  Four cameras with different discs into
    one batched, pipelined oat-track-hip; every stream must equal ITS oracle frame by frame, one token out per
    token in, each carrying its own frame's Sample."""
    import oracle_lib as O
    from oat_amd.synth import SyntheticStream
    rows, cols, n, ncam = 240, 320, 20, 4
    streams = [SyntheticStream(rows, cols, 20 + s, n_discs=1, radius=8 + 3 * s) for s in range(ncam)]
    frames = [[st.frame(t, with_discs=t > 0) for t in range(n)] for st in streams]
    got = _run_batched(host_bins, tmp_path, frames, ring=ring)
    p = O.hsv_params(h_lo=100, h_hi=125, s_lo=150, s_hi=256, v_lo=100, v_hi=256, erode=3, dilate=7,
                     min_area=20.0, max_area=1e5)
    areas = set()
    for s in range(ncam):
        assert len(got[s]) == n, (s, len(got[s]))
        orc = O.Mog2(rows, cols, 3)
        hits = 0
        for t, (f, g) in enumerate(zip(frames[s], got[s])):
            want, _ = O.chain_step(orc, f, 0.01, p)
            assert g["tick"] == t + 1 and g["usec"] == (t + 1) * 5000, (s, t, g)
            assert g["pos_ok"] == want["valid"], (s, t)
            if want["valid"]:
                hits += 1
                areas.add(want["a00"])
                assert abs(g["pos_xy"][0] - want["x"]) < 1e-4 and abs(g["pos_xy"][1] - want["y"]) < 1e-4, (s, t)
        assert hits >= n - 3, (s, hits)
    assert len(areas) >= ncam                     # the cameras really saw different things


@pytest.mark.gpu
def test_batched_tracker_free_running_cameras_publish_between_their_copies(host_bins, tmp_path):
    """Free-running cameras (frames always waiting) with a ring of four: the tracker hands finished result sets out SINK
    by SINK between the copies of the next round (oat_track_hip.cpp publish_some, r04) and pairs frames for the
    per-pixel kernel.  One token out per token in, in order, every camera equal to ITS oracle, each token with its
    frame's Sample."""
    import oracle_lib as O
    from oat_amd.synth import SyntheticStream
    rows, cols, n, ncam = 240, 320, 60, 5
    streams = [SyntheticStream(rows, cols, 50 + s, n_discs=1, radius=8 + 2 * s) for s in range(ncam)]
    frames = [[st.frame(t, with_discs=t > 0) for t in range(n)] for st in streams]
    got = _run_batched(host_bins, tmp_path, frames, ring=4, fps=0, extra=("--timing",))
    p = O.hsv_params(h_lo=100, h_hi=125, s_lo=150, s_hi=256, v_lo=100, v_hi=256, erode=3, dilate=7,
                     min_area=20.0, max_area=1e5)
    for s in range(ncam):
        assert len(got[s]) == n, (s, len(got[s]))
        orc = O.Mog2(rows, cols, 3)
        for t, (f, g) in enumerate(zip(frames[s], got[s])):
            want, _ = O.chain_step(orc, f, 0.01, p)
            assert g["tick"] == t + 1 and g["pos_ok"] == want["valid"], (s, t, g)
            if want["valid"]:
                assert abs(g["pos_xy"][0] - want["x"]) < 1e-4 and abs(g["pos_xy"][1] - want["y"]) < 1e-4, (s, t)


@pytest.mark.gpu
@pytest.mark.parametrize("ncam,devices", [(4, "0,0"), (3, "0,0"), (2, "0,0,0"), (16, "0,0,0,0,0,0,0,0"), (9, "0,0,0,0,0,0,0,0")])
def test_batched_tracker_sharded_over_device_contexts(host_bins, tmp_path, ncam, devices):
    """`oat-track-hip --gpu-index D0,D1,..`: the C++ launcher of SURVEY 8e's partition -- the SOURCE list cut into
    contiguous blocks, one batched tracker (own context, own thread) per listed device.  On a one-GPU box the same
    device is listed more than once: two or three contexts on device 0, even and uneven blocks, more devices than
    cameras.  Every camera must equal ITS oracle, token by token.  (BASELINE configs[3] / [4] are this with eight
    devices; unmeasured on multi-GPU hardware.)  r04: the EIGHT-shard forms on the one GPU -- 16 cameras as 8 contexts x 2,
    and 9 cameras (uneven: ceil(9/8) = 2 per shard, five shards, the last with one camera)."""
    import oracle_lib as O
    from oat_amd.synth import SyntheticStream
    rows, cols, n = 240, 320, 16
    streams = [SyntheticStream(rows, cols, 40 + s, n_discs=1, radius=8 + 3 * s) for s in range(ncam)]
    frames = [[st.frame(t, with_discs=t > 0) for t in range(n)] for st in streams]
    got = _run_batched(host_bins, tmp_path, frames, ring=2, extra=("--gpu-index", devices))
    p = O.hsv_params(h_lo=100, h_hi=125, s_lo=150, s_hi=256, v_lo=100, v_hi=256, erode=3, dilate=7,
                     min_area=20.0, max_area=1e5)
    for s in range(ncam):
        assert len(got[s]) == n, (s, len(got[s]))
        orc = O.Mog2(rows, cols, 3)
        hits = 0
        for t, (f, g) in enumerate(zip(frames[s], got[s])):
            want, _ = O.chain_step(orc, f, 0.01, p)
            assert g["tick"] == t + 1 and g["pos_ok"] == want["valid"], (s, t, g)
            if want["valid"]:
                hits += 1
                assert abs(g["pos_xy"][0] - want["x"]) < 1e-4 and abs(g["pos_xy"][1] - want["y"]) < 1e-4, (s, t)
        assert hits >= n - 3, (s, hits)


@pytest.mark.gpu
@pytest.mark.parametrize("ncam", [1, 3])
def test_ingest_root_scatter_tracker_world_of_one_matches_the_oracle(host_bins, tmp_path, ncam):
    """`oat-track-hip --ingest-root 0 --gpu-index 0`: the stream-to-rank scatter behind the C++ boundary --
    ONE process, ncclCommInitAll over the listed devices, per step every camera's frame into the root device's staging slot,
    ncclGroupStart / ncclSend x (N - 1) / ncclRecv / ncclGroupEnd on transfer streams, every shard's context ordered behind its
    transfer by an event, slot reuse gated by oatgpu_track_input_consumed (oat_amd/host/scatter_tracker.hpp).  Here the world
    of one there is hardware for: the communicator is created for real, the root's block is consumed in place, every camera
    equals ITS oracle token by token in order (PositionDetector.cpp:58-99), and --timing reports bytes per peer and ms per
    step.  N > 1 is the same code with the sends in it: unverified until a multi-GPU node runs it."""
    import oracle_lib as O
    from oat_amd.synth import SyntheticStream
    rows, cols, n = 240, 320, 24
    streams = [SyntheticStream(rows, cols, 70 + s, n_discs=1, radius=8 + 3 * s) for s in range(ncam)]
    frames = [[st.frame(t, with_discs=t > 0) for t in range(n)] for st in streams]
    err = tmp_path / "tracker.err"
    with open(err, "w") as ef:
        got = _run_batched(host_bins, tmp_path, frames, ring=2, extra=("--gpu-index", "0", "--ingest-root", "0", "--timing"), tracker_stderr=ef)
    p = O.hsv_params(h_lo=100, h_hi=125, s_lo=150, s_hi=256, v_lo=100, v_hi=256, erode=3, dilate=7, min_area=20.0, max_area=1e5)
    for s in range(ncam):
        assert len(got[s]) == n, (s, len(got[s]))
        orc = O.Mog2(rows, cols, 3)
        hits = 0
        for t, (f, g) in enumerate(zip(frames[s], got[s])):
            want, _ = O.chain_step(orc, f, 0.01, p)
            assert g["tick"] == t + 1 and g["pos_ok"] == want["valid"], (s, t, g)
            if want["valid"]:
                hits += 1
                assert abs(g["pos_xy"][0] - want["x"]) < 1e-4 and abs(g["pos_xy"][1] - want["y"]) < 1e-4, (s, t)
        assert hits >= n - 3, (s, hits)
    text = open(err).read()
    assert "track-scatter[" in text and "RCCL" in text and f"{ncam * rows * cols * 3} bytes per peer and step" in text, text[-600:]
    assert f"{n} steps, {ncam} cameras over 1 device(s)" in text, text[-600:]


def _start_batched(host_bins, tmp_path, streams_frames, extra=(), ring=2, fps=200):
    """Like _run_batched, but the cameras may differ in geometry and frame count and nothing is waited for: returns
    (tracker, readers, reader files, feeders, addresses, t_feeders_started)."""
    n = len(streams_frames)
    tag = "oat_t_" + uuid.uuid4().hex[:8]
    srcs = [f"{tag}raw{s}" for s in range(n)]
    snks = [f"{tag}pos{s}" for s in range(n)]
    B = lambda b: os.path.join(host_bins, b)
    files = [open(tmp_path / f"reader{s}.out", "w+") for s in range(n)]
    readers = [subprocess.Popen([B("oat-posi-cout"), a], stdout=f, stderr=subprocess.DEVNULL, text=True) for a, f in zip(snks, files)]
    tracker = subprocess.Popen([B("oat-track-hip"), ",".join(srcs), ",".join(snks), "-a", "0.01", "--area", "[20,100000]",
                                "-H", "[100,125]", "-S", "[150,256]", "-V", "[100,256]", "-e", "3", "-d", "7",
                                "--ring", str(ring)] + list(extra), stderr=subprocess.PIPE, text=True)
    _consumers_ready(*srcs, *snks)
    feeders = []
    t0 = time.monotonic()
    for s in range(n):
        rows, cols = streams_frames[s][0].shape[:2]
        raw = tmp_path / f"frames{s}.raw"
        np.stack(streams_frames[s]).tofile(raw)
        feeders.append(subprocess.Popen([B("oat-frameserve-raw"), srcs[s], "-f", str(raw), "--rows", str(rows), "--cols",
                                         str(cols), "-n", str(len(streams_frames[s])), "-r", str(fps)]))
    return tracker, readers, files, feeders, srcs + snks, t0


def _tokens(fl):
    fl.seek(0)
    return [json.loads(l) for l in fl.read().splitlines() if l.strip()]


@pytest.mark.gpu
def test_failing_shard_ends_its_sinks_at_once_and_the_others_keep_their_tokens(host_bins, tmp_path):
    """VERDICT r03 item 5 / weak 11.  Two contexts on one GPU (`--gpu-index 0,0`, four cameras): camera 3 has another
    geometry than camera 2, so shard 1 fails while connecting.  Its tracker is destroyed at once by its own thread --
    its SINKs are bound on the way out and go END (lib/shmemdf/Sink.h:73-91), so the consumers of cameras 2 and 3 end
    right away instead of blocking for as long as shard 0 runs -- shard 0's cameras get every one of their tokens, equal
    to their oracles, and the process exit code still reports the failure (framefilter/main.cpp:278-295)."""
    import oracle_lib as O
    from oat_amd.synth import SyntheticStream
    rows, cols, n = 240, 320, 50
    streams = [SyntheticStream(rows, cols, 60 + s, n_discs=1, radius=8 + 3 * s) for s in range(3)]
    frames = [[st.frame(t, with_discs=t > 0) for t in range(n)] for st in streams]
    odd = SyntheticStream(120, 160, 63, n_discs=1, radius=6)
    frames.append([odd.frame(t, with_discs=t > 0) for t in range(n)])
    tracker, readers, files, feeders, addrs, t0 = _start_batched(host_bins, tmp_path, frames, extra=("--gpu-index", "0,0"), fps=10)
    try:
        # shard 1's consumers end while shard 0 (50 frames at 10 fps = 5 s) is still being served (the margin is for a slow
        # box's process start-up, not for the component: the SINKs go END as soon as the shard's connect fails)
        for r in readers[2:]:
            r.wait(timeout=4.0)
        t_end = time.monotonic() - t0
        assert t_end < 3.0, t_end
        assert readers[0].poll() is None and readers[1].poll() is None and tracker.poll() is None      # shard 0 goes on
        for r in readers[:2]:
            r.wait(timeout=120)
        for f in feeders:
            f.wait(timeout=60)
        _, err = tracker.communicate(timeout=60)
    finally:
        for p_ in readers + feeders + [tracker]:
            if p_.poll() is None:
                p_.kill()
        subprocess.run([os.path.join(host_bins, "oat-clean-hip")] + addrs, capture_output=True)
    assert tracker.returncode == 255 and "same frame geometry" in err, (tracker.returncode, err[-500:])
    assert _tokens(files[2]) == [] and _tokens(files[3]) == []
    p = O.hsv_params(h_lo=100, h_hi=125, s_lo=150, s_hi=256, v_lo=100, v_hi=256, erode=3, dilate=7, min_area=20.0, max_area=1e5)
    for s in range(2):
        got = _tokens(files[s])
        assert len(got) == n, (s, len(got))
        orc = O.Mog2(rows, cols, 3)
        for t, (f, g) in enumerate(zip(frames[s], got)):
            want, _ = O.chain_step(orc, f, 0.01, p)
            assert g["tick"] == t + 1 and g["pos_ok"] == want["valid"], (s, t, g)
            if want["valid"]:
                assert abs(g["pos_xy"][0] - want["x"]) < 1e-4 and abs(g["pos_xy"][1] - want["y"]) < 1e-4, (s, t)


@pytest.mark.gpu
def test_end_on_one_camera_in_the_middle_of_a_round(host_bins, tmp_path):
    """END on camera 1 of 3 (its frame server stops 5 frames early) while camera 0's frame of that round is already
    staged: the partly staged set is given up (oatgpu_track_stage_abort), every COMPLETE round still gets its token
    on all three SINKs -- equal to the oracles -- and the component exits 0 like the reference at END of stream
    (framefilter/main.cpp:271-276)."""
    import oracle_lib as O
    from oat_amd.synth import SyntheticStream
    rows, cols, n, short = 240, 320, 20, 15
    streams = [SyntheticStream(rows, cols, 70 + s, n_discs=1, radius=8 + 3 * s) for s in range(3)]
    frames = [[st.frame(t, with_discs=t > 0) for t in range(n if s != 1 else short)] for s, st in enumerate(streams)]
    tracker, readers, files, feeders, addrs, _ = _start_batched(host_bins, tmp_path, frames, fps=100)
    try:
        for r in readers:
            r.wait(timeout=120)
        _, err = tracker.communicate(timeout=60)
        feeders[1].wait(timeout=60)
    finally:
        for p_ in readers + feeders + [tracker]:                     # (cameras 0 and 2 still hold frames nobody will take)
            if p_.poll() is None:
                p_.kill()
        subprocess.run([os.path.join(host_bins, "oat-clean-hip")] + addrs, capture_output=True)
    assert tracker.returncode == 0, err[-500:]
    p = O.hsv_params(h_lo=100, h_hi=125, s_lo=150, s_hi=256, v_lo=100, v_hi=256, erode=3, dilate=7, min_area=20.0, max_area=1e5)
    for s in range(3):
        got = _tokens(files[s])
        assert len(got) == short, (s, len(got))
        orc = O.Mog2(rows, cols, 3)
        for t, (f, g) in enumerate(zip(frames[s], got)):
            want, _ = O.chain_step(orc, f, 0.01, p)
            assert g["tick"] == t + 1 and g["pos_ok"] == want["valid"], (s, t, g)
            if want["valid"]:
                assert abs(g["pos_xy"][0] - want["x"]) < 1e-4 and abs(g["pos_xy"][1] - want["y"]) < 1e-4, (s, t)


def _write_pnm(path, img):
    img = np.ascontiguousarray(img, np.uint8)
    if img.ndim == 2:
        hdr = f"P5\n# written by the tests\n{img.shape[1]} {img.shape[0]}\n255\n".encode()
        data = img.tobytes()
    else:
        hdr = f"P6\n{img.shape[1]} {img.shape[0]}\n255\n".encode()
        data = img[..., ::-1].tobytes()                 # PPM is RGB
    with open(path, "wb") as f:
        f.write(hdr + data)


def _grey_chain(host_bins, tmp_path, frames, filt_args, color="GREY"):
    """frameserve -> oat-framefilt-hip <filt_args> -> oat-posidet-hip thresh -> posi-cout (GREY frames)."""
    rows, cols = frames[0].shape[:2]
    raw = tmp_path / "g.raw"
    np.stack(frames).tofile(raw)
    tag = "oat_t_" + uuid.uuid4().hex[:8]
    a_raw, a_f, a_pos = tag + "raw", tag + "f", tag + "pos"
    B = lambda b: os.path.join(host_bins, b)
    reader = subprocess.Popen([B("oat-posi-cout"), a_pos], stdout=subprocess.PIPE, text=True)
    procs = [subprocess.Popen([B("oat-posidet-hip"), "thresh", a_f, a_pos, "-T", "[100,256]", "-a", "[4,100000]"]),
             subprocess.Popen([B("oat-framefilt-hip"), filt_args[0], a_raw, a_f] + list(filt_args[1:]))]
    _consumers_ready(a_pos, a_raw, a_f)
    feeder = subprocess.Popen([B("oat-frameserve-raw"), a_raw, "-f", str(raw), "--rows", str(rows), "--cols", str(cols),
                               "-C", color, "-n", str(len(frames)), "-r", "200"])
    try:
        out, _ = reader.communicate(timeout=120)
        feeder.wait(timeout=60)
        for p in procs:
            p.wait(timeout=60)
    finally:
        for p in procs + [feeder, reader]:
            if p.poll() is None:
                p.kill()
        subprocess.run([B("oat-clean-hip"), a_raw, a_f, a_pos], capture_output=True)
    assert all(p.returncode == 0 for p in procs), [p.returncode for p in procs]
    return [json.loads(l) for l in out.splitlines() if l.strip()]


@pytest.mark.gpu
def test_framefilt_mask_and_bsub_read_pnm_files(host_bins, tmp_path):
    """`framefilt mask -f FILE` (FrameMasker.cpp:45-75) and `framefilt bsub -f FILE` (BackgroundSubtractor.cpp:52-100)
    from the binaries, with Netpbm files standing in for cv::imread: two bright squares, the mask / the
    background image removes the left one, the detector must report the right one's centroid."""
    import oracle_lib as O
    rows, cols = 96, 128
    f = np.zeros((rows, cols), np.uint8)
    f[20:40, 10:30] = 200           # left square  (area 400)
    f[50:70, 80:110] = 220          # right square (area 600)
    frames = [f.copy() for _ in range(4)]
    p = O.hsv_params(h_lo=100, h_hi=256, erode=0, dilate=0, min_area=4.0, max_area=1e5)
    # mask: keep only the right half
    m = np.zeros((rows, cols), np.uint8)
    m[:, 64:] = 255
    _write_pnm(tmp_path / "m.pgm", m)
    got = _grey_chain(host_bins, tmp_path, frames, ["mask", "-f", str(tmp_path / "m.pgm")])
    want, _ = O.detect_thresh(np.where(m > 0, f, 0).astype(np.uint8), p)
    assert len(got) == 4 and want["valid"]
    for g in got:
        assert g["pos_ok"] and abs(g["pos_xy"][0] - want["x"]) < 1e-4 and abs(g["pos_xy"][1] - want["y"]) < 1e-4
    # bsub: background file holds the left square only -> saturating subtraction leaves the right one
    bg = np.zeros((rows, cols), np.uint8)
    bg[20:40, 10:30] = 200
    _write_pnm(tmp_path / "bg.pgm", bg)
    got = _grey_chain(host_bins, tmp_path, frames, ["bsub", "-f", str(tmp_path / "bg.pgm")])
    want, _ = O.detect_thresh(np.clip(f.astype(int) - bg, 0, 255).astype(np.uint8), p)
    assert len(got) == 4 and want["valid"]
    for g in got:
        assert g["pos_ok"] and abs(g["pos_xy"][0] - want["x"]) < 1e-4 and abs(g["pos_xy"][1] - want["y"]) < 1e-4
