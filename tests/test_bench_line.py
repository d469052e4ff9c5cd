"""bench.py's ONE stdout line (a 21 KB line was not parsed by the driver): built from canned result dicts, held to
4 KB, parseable, carrying the contract's keys -- and `--gpus N` as an N-rank run: the errors a box without the devices gets."""
import copy
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "frac_one_frame", "frac_benched", "bytes_per_launch",
            "avg_launch_ms", "waste_ratio", "measured_stream_copy_GBps")
CPU = ("value", "unit", "cores", "kind", "sample", "method", "value_1thread", "nproc", "cpu_model")


def _full():
    """Everything a default N = 1 run measured (the full line of a real run on an MI355X, 21 KB)."""
    with open(os.path.join(ROOT, "tests", "golden", "bench_full_r04_sample.json")) as f:
        return json.load(f)


def _eight_ranks(full):
    d = copy.deepcopy(full)
    d["n_gpus"] = 8
    d["cpu_baseline"] = None
    d["pipeline"] = None
    # N > 1 (r08): the other north-star size on every rank, and the benched launch priced from the slowest rank's HIP events
    d["extra_workloads"] = {"1080p8": dict(name="1080p8", value=612345.678, unit="frames/s", steps=20, blocks=9, parity="ok",
                                           k_mog_fused_ms=0.2212345, k_mog_fused_ms_min=0.2112345, streams_total=64,
                                           per_rank=[dict(rank=r, parity="ok", found=1000, k_mog_fused_ms=0.22) for r in range(8)])}
    d["roofline"] = {"bound": "hbm", "kernel": "k_mog_fused", "peak": 8000.0, "unit": "GB/s", "achieved": None, "frac": None, "traffic": None,
                     "frac_benched": 0.61234567, "frac_benched_source": "audit", "k_mog_fused_ms_ranks": {"min": 0.0991234, "max": 0.1012345},
                     "benched_workload": {"avg_launch_ms": 0.1012345, "traffic": 496123456.0}}
    d["value_mean"] = d["value"] * 0.99
    d["positions_with_target"] = 11600
    d["partition"] = {"rule": "x" * 150, "streams_total": 64,
                      "per_rank": [dict(rank=r, device=r, streams=[8 * r, 8 * r + 8], parity="ok", positions_found=12345,
                                        block_ms=10.123456789, k_mog_fused_ms=0.20123456, ms_per_step_local=0.1012345) for r in range(8)]}
    d["rccl"] = {"ranks": 8, "backend": "nccl", "version": "2.26.6"}
    d["scatter_ingest"] = dict(fps=123456.789, ms_per_step=0.51234567, steps=192, depth=2, backend="nccl", bytes_per_peer=49766400,
                               parity="ok", per_rank=[dict(rank=r, parity="ok", found=100) for r in range(8)], what="y" * 400,
                               also=dict(workload="1080p8", fps=423456.789, ms_per_step=0.15123456, steps=96, depth=2, backend="nccl",
                                         bytes_per_peer=49766400, parity="ok", per_rank=[dict(rank=r, parity="ok") for r in range(8)], what="y" * 400))
    d["latency_us"] = dict(saturated_p50=415.123456, saturated_p99=460.1, ring_depth=8, single_p50=180.2, single_p99=199.9)
    return d


@pytest.mark.parametrize("shape", ["solo", "eight"])
def test_slim_line_is_small_parseable_and_complete(shape):
    full = _full() if shape == "solo" else _eight_ranks(_full())
    assert len(json.dumps(full)) > (8000 if shape == "solo" else 5000)      # the canned dict really is a big one
    text = json.dumps(bench.slim_line(full), separators=(",", ":"))
    assert len(text) <= bench.SLIM_LIMIT, len(text)
    assert "\n" not in text
    j = json.loads(text)
    for k in CONTRACT:
        assert k in j, k
    assert j["value"] == pytest.approx(full["value"], rel=1e-6) and j["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-6)
    assert j["n_gpus"] == full["n_gpus"] and j["steps"] == full["steps"] and j["warmup"] == full["warmup"]
    assert j["config"]["workload"] == "4k1" and "model" not in j["config"]
    for k in ROOFLINE:
        assert k in j["roofline"], k
    if shape == "solo":
        assert j["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-3)
        assert j["roofline"]["achieved"] / j["roofline"]["peak"] == pytest.approx(j["roofline"]["frac"], rel=1e-3)
        for k in CPU:
            assert j["cpu_baseline"][k] is not None, k
        assert j["cpu_baseline"]["value"] == pytest.approx(full["cpu_baseline"]["value"], rel=1e-4)
        assert set(j["extra_workloads"]) == {"1080p16", "1080p8", "1080p1", "vga1"} and j["extra_parity"] == "ok"
        assert all(isinstance(v, float) for v in j["extra_workloads"].values())
        assert len(j["pipeline"]) == 5
    else:
        assert j["cpu_baseline"] is None and j["roofline"]["frac"] is None
        assert [r[:5] for r in j["partition"]["ranks"]] == [[r, r, 8 * r, 8 * r + 8, "ok"] for r in range(8)]
        assert j["scatter_ingest"]["parity"] == "ok" and j["scatter_ingest"]["bytes_per_peer"] == 49766400
        assert "what" not in j["scatter_ingest"] and "per_rank" not in j["scatter_ingest"]
        assert j["rccl"]["ranks"] == 8
        # the whole N > 1 record in the one line
        assert j["extra_workloads"] == {"1080p8": pytest.approx(612345.678, rel=1e-5)} and j["extra_parity"] == "ok"
        assert j["roofline"]["frac_benched"] == pytest.approx(0.6123, rel=1e-3) and j["roofline"]["frac_benched_source"] == "audit"
        assert j["roofline"]["k1_ms_ranks"] == [pytest.approx(0.09912, rel=1e-3), pytest.approx(0.1012, rel=1e-3)]
        assert j["roofline"]["benched_launch_ms"] == pytest.approx(0.10123, rel=1e-3)
        also = j["scatter_ingest"]["also"]
        assert also["workload"] == "1080p8" and also["parity"] == "ok" and "what" not in also and "per_rank" not in also
        assert j["value_mean"] == pytest.approx(full["value"] * 0.99, rel=1e-5) and j["positions_with_target"] == 11600
    # no prose: nothing in the line is a long string
    def longest(o):
        if isinstance(o, str):
            return len(o)
        if isinstance(o, dict):
            return max([longest(v) for v in o.values()] + [0])
        if isinstance(o, list):
            return max([longest(v) for v in o] + [0])
        return 0
    assert longest(j) <= 110


def test_emit_prints_one_line_and_keeps_the_detail(tmp_path, capsys, monkeypatch):
    monkeypatch.setattr(bench, "DETAIL_PATH", str(tmp_path / "bench_detail.json"))
    full = _full()
    bench.emit(full)
    cap = capsys.readouterr()
    lines = [l for l in cap.out.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) <= bench.SLIM_LIMIT
    assert json.loads(lines[-1])["value"] == pytest.approx(full["value"], rel=1e-6)
    with open(tmp_path / "bench_detail.json") as f:
        assert json.load(f)["roofline"]["fractions"] == full["roofline"]["fractions"]       # the prose lives here
    assert "bench detail:" in cap.err


def test_emit_never_exceeds_the_limit_even_with_oversized_blocks(tmp_path, capsys, monkeypatch):
    monkeypatch.setattr(bench, "DETAIL_PATH", str(tmp_path / "d.json"))
    full = _eight_ranks(_full())
    full["partition"]["per_rank"] = [dict(rank=r, device=r, streams=[r, r + 1], parity="rank %d: " % r + "z" * 90) for r in range(64)]
    text = bench.emit(full)
    capsys.readouterr()
    assert len(text) <= bench.SLIM_LIMIT and json.loads(text)["value"] == pytest.approx(full["value"], rel=1e-6)


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=300)


def test_gpus_n_without_the_devices_is_an_error_not_a_silent_single_gpu_run():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this box has the devices")
    r = _run(["--gpus", "2", "--quick", "--steps", "5"])
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    assert "needs 2 visible device(s)" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_gpus_must_match_the_launchers_world():
    r = _run(["--gpus", "2", "--quick", "--steps", "5"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "refusing to run a mislabelled job" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_a_hanging_scatter_leg_costs_the_run_that_block_not_its_line(tmp_path):
    """bench.scatter_with_watchdog: the scatter leg runs last; should its transport hang, rank 0 prints the line it already has
    (scatter_ingest marked "timeout") and the rank ends with rc 0 instead of sitting in a collective until the job is killed."""
    code = r"""
import json, sys, time, types
sys.path.insert(0, %r)
import bench
bench.DETAIL_PATH = %r
bench.scatter_leg = lambda *a, **k: time.sleep(30)
full = json.load(open(%r))
full["n_gpus"] = 2
args = types.SimpleNamespace(scatter_timeout=0.5, workload="vga1", backend="gloo", scatter_steps=8)
bench.scatter_with_watchdog(args, 2, 0, None, None, full, time.perf_counter())
print("NOT REACHED")
""" % (ROOT, str(tmp_path / "d.json"), os.path.join(ROOT, "tests", "golden", "bench_full_r04_sample.json"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and "NOT REACHED" not in r.stdout, r.stdout[-500:]
    j = json.loads(lines[0])
    assert j["scatter_ingest"]["parity"] == "timeout" and j["n_gpus"] == 2 and j["value"] > 0
    assert "given up" in r.stderr


def test_a_hanging_second_workload_costs_the_run_that_block_not_its_line(tmp_path):
    """Round 6: at N > 1 the second workload's leg (extra_workloads.1080p8) runs with the scatter legs, LAST and under their
    watchdog -- the driver's multi-GPU run is a one-shot and `value` must not depend on a leg that comes after it.  A leg
    that hangs (a rank that died in a collective): rank 0 prints the line it has, the missing blocks marked, rc 0."""
    code = r"""
import json, sys, time, types
sys.path.insert(0, %r)
import bench
bench.DETAIL_PATH = %r
full = json.load(open(%r))
full["n_gpus"] = 8
full["extra_workloads"] = None
full["scatter_ingest"] = None
args = types.SimpleNamespace(scatter_timeout=0.3, extra_timeout=0.2, workload="4k1", backend="gloo", scatter_steps=8)
bench.scatter_with_watchdog(args, 8, 0, None, None, full, time.perf_counter(), also="1080p8", extra_fn=lambda: time.sleep(30))
print("NOT REACHED")
""" % (ROOT, str(tmp_path / "d.json"), os.path.join(ROOT, "tests", "golden", "bench_full_r04_sample.json"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and "NOT REACHED" not in r.stdout, r.stdout[-500:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["value"] > 0 and j["scatter_ingest"]["parity"] == "timeout"
    assert j["extra_workloads"] == {"1080p8": None} and j["extra_parity"] == {"1080p8": "timeout"}
    assert "given up" in r.stderr


def test_slim_line_survives_a_run_that_measured_almost_nothing():
    """A leg that failed leaves None behind (the probes of bench.py never break the line): the builder takes any of it."""
    full = {"metric": "m", "value": 1.0, "unit": "frames/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1000.0,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"name": "vga1"}, "roofline": {"frac": None}, "cpu_baseline": None, "stage_ms": None, "latency_us": None,
            "extra_workloads": {}, "pipeline": {"error": "x"}, "partition": None, "scatter_ingest": {"error": "boom", "parity": "error"}}
    j = json.loads(json.dumps(bench.slim_line(full)))
    assert j["value"] == 1.0 and j["roofline"]["frac"] is None and j["cpu_baseline"] is None
    assert j["scatter_ingest"] == {"error": "boom", "parity": "error"} and j["partition"]["ranks"] == []
    assert float("nan") != float("nan") and bench._sig(float("nan")) is None and bench._sig(float("inf")) is None
