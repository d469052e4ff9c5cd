"""Child of tests/test_gpu_parity.py::test_early_order_survives_a_tool_that_serialises_dispatches.

The default path of a context of ONE stream of >= 4 MP (and of two 1080p streams) parks a blob workgroup that waits, on the
device, for a ticket a LATER kernel of another stream publishes -- it relies on two kernels being resident at once, which HIP
does not promise.  Run under something that serialises kernel dispatches (AMD_SERIALIZE_KERNEL=3, HIP_LAUNCH_BLOCKING=1, a
counter-collecting rocprofv3) the parked workgroup may wait for a kernel that cannot start: after 100 ms it declines its
frame to the global kernels and the context stops parking (oatgpu_early_blob_timeouts, oatgpu_last_error).

This process runs `frames` frames through the pipelined device-frame path with the ring kept full, every result against
the oracle (PositionDetector.cpp:58-99: one token out per token in, in order), then masks and the whole model, and prints one
JSON line.  argv: rows cols n_streams frames [frames_per_launch: 2 = oatgpu_set_fusion(2), both frames of a step park together]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import oracle_lib as O  # noqa: E402
import oat_amd as A  # noqa: E402


def main():
    rows, cols, n, T = (int(x) for x in sys.argv[1:5])
    fusion = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    ring = 4
    win = dict(h_thresh=(100, 125), s_thresh=(150, 256), v_thresh=(100, 256))
    hp = A.HotPath(rows, cols, n_streams=n, ring_depth=ring, adaptation_coeff=0.01, erode=3, dilate=5, area=(20.0, 1e6), **win)
    if fusion == 2:
        hp.set_fusion(2)
    p = O.hsv_params(h_lo=100, h_hi=125, s_lo=150, s_hi=256, v_lo=100, v_hi=256, erode=3, dilate=5, min_area=20.0, max_area=1e6)
    orc = [O.Mog2(rows, cols, 3) for _ in range(n)]
    rng = np.random.default_rng(77)
    base = rng.integers(90, 150, (n, rows, cols, 3)).astype(np.int16)

    def frame(t):
        f = np.clip(base + rng.integers(-5, 6, base.shape), 0, 255).astype(np.uint8)
        if t > 0:
            for s in range(n):
                cy, cx = 40 + (7 * t + 30 * s) % (rows - 100), 50 + (11 * t + 40 * s) % (cols - 120)
                f[s, cy:cy + 30, cx:cx + 45] = (255, 64, 0)
        return f
    frames = [frame(t) for t in range(T)]
    dev = [torch.from_numpy(f).cuda() for f in frames]
    torch.cuda.synchronize()
    got, shapes = [], []
    t0 = time.perf_counter()
    for d in dev:
        hp.enqueue_dev(d.data_ptr(), keepalive=d)
        shapes.append(hp.last_step_shape())
        if hp.outstanding() >= ring:
            got.append(hp.collect())
    while hp.outstanding():
        got.append(hp.collect())
    wall = time.perf_counter() - t0
    bad = []
    for t, f in enumerate(frames):
        for s in range(n):
            want = O.chain_step(orc[s], f[s], 0.01, p)[0]
            g = got[t][s]
            same = g.position_valid == want["valid"] and g.area == want["area"] and (
                not want["valid"] or ((g.a00, g.a10, g.a01) == (want["a00"], want["a10"], want["a01"]) and
                                      g.first_pixel == want["first_pixel"] and g.x == want["x"] and g.y == want["y"]))
            if not same:
                bad.append((t, s))
    model_ok = True
    for s in range(n):
        nm, w, v, m, _ = hp.mog_state(s)
        nm_o, w_o, v_o, m_o = orc[s].state()
        live = np.arange(w_o.shape[1])[None, :] < nm_o[:, None]
        model_ok = model_ok and bool((nm == nm_o).all() and (w[live] == w_o[live]).all() and (v[live] == v_o[live]).all() and
                                     (m[live] == m_o[live]).all())
    err = hp.lib.oatgpu_last_error(hp.ctx)
    print(json.dumps(dict(results=len(got), mismatches=bad[:8], model_ok=model_ok, timeouts=hp.early_blob_timeouts(), wall_s=wall,
                          early_steps=sum(1 for _, e in shapes if e), steps=len(shapes), last_error=(err.decode() if err else ""), fusion=fusion,
                          env={k: os.environ.get(k) for k in ("AMD_SERIALIZE_KERNEL", "HIP_LAUNCH_BLOCKING")})))
    hp.close()


if __name__ == "__main__":
    main()
