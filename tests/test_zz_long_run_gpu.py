"""Long-run whole-model parity at real frame sizes (tools/state_check.py), last in the GPU suite: the pipelined path --
two frames a launch -- for a hundred frames and more, then further frames through the traffic-audit instantiations of
the per-pixel kernel; every position and the WHOLE model against the oracle, bit for bit."""
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import oat_amd
    return oat_amd


def test_long_run_model_parity_with_audited_steps(A):
    """The pipelined path at a real frame size, full occupancy: 120 frames of one 1080p SURVEY-8d stream through
    enqueue / collect (ring 4), then six more through the traffic-audit instantiation of the per-pixel kernel; every
    position and the WHOLE model (counters, weights, variances, means) must be the oracle's, bit for bit.  (The
    short model-parity sequences of test_gpu_parity.py run small frames; a fault of round 2 -- the audit instantiation updating the
    model wrongly after a change of a load's type -- showed only here, tools/state_check.py.)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import state_check
    msgs = []
    assert state_check.run(1080, 1920, 120, 24, audited=6, log=msgs.append) == 0, msgs
    assert state_check.run(480, 640, 90, 16, streams=3, audited=4, log=msgs.append) == 0, msgs
    # a dense model (five live modes everywhere): from its second density probe on the library runs the instantiation
    # whose slot-1..4 loads use the streaming cache policy -- same numbers
    assert state_check.run(480, 640, 100, 10, streams=2, audited=4, dense=True, log=msgs.append) == 0, msgs


def _tools():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_full_size_default_path_whole_model(A):
    """What bench.py's `value` runs -- the pipelined path with two frames a launch -- at BASELINE's full sizes, whole
    model against the oracle: one 4K stream (configs[4]'s per-GPU shard; 61 frames + 4 audited) and 16 batched 1080p
    streams (configs[2]; 40 frames, every stream against its own oracle)."""
    _tools()
    import state_check
    msgs = []
    assert state_check.run(2160, 3840, 61, 12, audited=4, log=msgs.append) == 0, msgs
    assert state_check.run(1080, 1920, 40, 12, streams=16, audited=0, log=msgs.append) == 0, msgs


def test_full_size_frozen_model(A):
    """Oat's default learning rate (framefilt mog without -a: adaptation_coeff 0) at full HD through the pipelined path: the
    FROZEN instantiations of the per-pixel kernel (records are stored only when their bits changed), two frames a launch
    and one, whole model and every position against the oracle -- what bench.py's value_default_learning_rate_0 runs."""
    _tools()
    import state_check
    msgs = []
    assert state_check.run(1080, 1920, 41, 12, alpha=0.0, audited=0, log=msgs.append) == 0, msgs
    assert state_check.run(1080, 1920, 21, 12, alpha=0.0, audited=0, fusion=1, log=msgs.append) == 0, msgs


def test_single_launch_audited_equals_product(A):
    """The round-2 fault, isolated (tools/k1_fault_probe.py): from one exported model, ONE audited launch must leave
    exactly what ONE product launch leaves (and what the oracle leaves), with two frames a launch and with one,
    four times over.  Round 2's audited two-frame instantiation failed this on 16 lanes of ~3 000 waves per launch --
    the gfx950 wide-store data hazard behind st_rec."""
    _tools()
    import k1_fault_probe
    msgs = []
    assert k1_fault_probe.run(1080, 1920, 40, 4, log=msgs.append) == 0, msgs


def _model_hash(hp, streams):
    import hashlib
    h = hashlib.sha256()
    for s in range(streams):
        for a in hp.mog_state(s)[:4]:
            h.update(a.tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("kind", ["bgr_fused", "bgr_one_frame", "bgr_dense_streaming_loads", "grey_fused", "bgr_audited"])
def test_whole_model_determinism_4k(A, kind):
    """Three runs of the same 4K sequence through every instantiation of the per-pixel kernel the product can launch
    (two frames a launch, one frame a launch, the streaming-load one a dense model switches to, GREY, and the audited
    one): the whole model must hash the same every time.  A run-to-run difference is how the wide-store hazard showed."""
    import numpy as np
    from oat_amd.synth import SyntheticStream, disc_hsv_window
    rows, cols = 2160, 3840
    dense = kind == "bgr_dense_streaming_loads"
    grey = kind == "grey_fused"
    n = 40 if dense else 14
    rng = np.random.default_rng(7)
    if dense:
        table = np.array([[20, 30, 40], [90, 200, 60], [200, 60, 120], [240, 240, 230], [40, 130, 220]], np.int16)
        phase = rng.integers(0, 5, (rows, cols))
        fr = [np.clip(table[(phase + t) % 5] + rng.integers(-5, 6, (rows, cols, 3), dtype=np.int16), 0, 255).astype(np.uint8)
              for t in range(5)]
    else:
        st = SyntheticStream(rows, cols, 0, n_discs=2)
        fr = [st.frame(9 * t, with_discs=t > 0) for t in range(6)]
        if grey:
            fr = [np.ascontiguousarray(f[:, :, 1]) for f in fr]
    hashes = []
    for rep in range(3):
        if grey:
            hp = A.HotPath(rows, cols, n_streams=1, adaptation_coeff=0.01, erode=3, dilate=7, area=(20.0, 1e5),
                           ring_depth=4, channels=1, h_thresh=(100, 256))
        else:
            hp = A.HotPath(rows, cols, n_streams=1, adaptation_coeff=0.01, erode=3, dilate=7, area=(20.0, 1e5),
                           ring_depth=4, **disc_hsv_window())
        if kind == "bgr_one_frame":
            hp.set_fusion(1)
        for t in range(n):
            if kind == "bgr_audited" and t == 4:
                while hp.outstanding():
                    hp.collect()
                hp.traffic_audit(True)
            hp.enqueue([fr[t % len(fr)]])
            if hp.outstanding() >= 4:
                hp.collect()
        while hp.outstanding():
            hp.collect()
        if kind == "bgr_audited":
            assert hp.traffic_read()["launches"] > 0
            hp.traffic_audit(False)
        hashes.append(_model_hash(hp, 1))
        hp.close()
    assert hashes[0] == hashes[1] == hashes[2], (kind, hashes)
