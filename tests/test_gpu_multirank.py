"""Multi-GPU readiness on ONE GPU: the N > 1 control flow of bench.py and an RCCL
import/initialisation check, so that a problem shows up before the driver's 8-GPU SCALE run.  What these
cannot show is scaling: that stays "unmeasured on hardware" until a SCALE record exists."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_over_gloo_share_one_gpu():
    """`torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 --backend gloo`: both ranks run the hot path on
    the one GPU, barriers and max-over-ranks timing work, rank 0 alone prints one line whose value counts BOTH
    ranks' streams, and every position of both ranks was found."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--workload", "vga1", "--steps", "60", "--warmup", "10", "--check-steps", "8"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 60 and j["scaling"] == "weak"
    assert j["positions_expected"] == 2 * 60 and j["positions_found"] >= 2 * 60 - 4
    assert abs(j["value"] - 2 * 60 / (j["ms_per_step"] * 60 / 1e3)) < 1e-6 * j["value"]
    assert j["parity"] == "ok"
    assert j["cpu_baseline"] is None and j["roofline"]["frac"] is None      # N = 1-only legs are skipped, and say so


def test_rccl_single_rank_init_allreduce_teardown():
    """backend "nccl" IS RCCL on ROCm: a world of one must initialise, reduce on the GPU and tear down."""
    code = r"""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.arange(8, dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
assert t.tolist() == list(range(8))
import oat_amd.dist as od
blk = od.scatter_frames(torch.arange(2 * 24, dtype=torch.uint8).view(2, 2, 4, 3), 2, (2, 4, 3), torch.device("cuda", 0))
assert blk.shape == (2, 2, 4, 3) and int(blk.sum()) == sum(range(48))
pipe = od.FrameScatterPipe(2, (2, 4, 3), torch.device("cuda", 0))
pipe.post(0, torch.full((2, 2, 4, 3), 7, dtype=torch.uint8))
assert int(pipe.take(0).sum()) == 7 * 48
dist.destroy_process_group()
print("rccl ok")
"""
    env = dict(os.environ, MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stderr[-3000:]
